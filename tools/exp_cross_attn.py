import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adt_str_amd import kernels as K
dev = torch.device("cuda:0")
B, H, S, T, d = 64, 6, 986, 128, 768
scale = 128 ** -0.5
torch.manual_seed(0)
qc = torch.randn((B * T, d), device=dev).bfloat16()
kvc = torch.randn((B * S, 2 * d), device=dev).bfloat16()
oc, lsec = K.attn_fwd(qc, kvc[:, :d], kvc[:, d:], B, H, T, S, scale, drop=(0.1, 7))
doc = torch.randn((B * T, d), device=dev).bfloat16()
dqc, dkvc = torch.empty_like(qc), torch.empty_like(kvc)
bg = torch.empty(3 * d, device=dev)
for _ in range(30):
    K.attn_fwd(qc, kvc[:, :d], kvc[:, d:], B, H, T, S, scale, drop=(0.1, 7))
    K.attn_bwd(qc, kvc[:, :d], kvc[:, d:], oc, doc, lsec, dqc, dkvc[:, :d], dkvc[:, d:], B, H, T, S, scale, drop=(0.1, 7), bias_grad=bg)
torch.cuda.synchronize()
