import torch
dev = "cuda:0"
for M, N, K in [(63104, 3072, 768), (63104, 768, 3072), (8192, 8192, 8192)]:
    a = torch.randn(M, K, device=dev).bfloat16(); w = torch.randn(N, K, device=dev).bfloat16()
    for _ in range(3):
        torch.matmul(a, w.t())
torch.cuda.synchronize()
