import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adt_str_amd import kernels as K
dev = torch.device("cuda:0")
def t(fn, n=200, warm=50):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
M = 8192
for N, Kd in ((2304, 768), (768, 768), (3072, 768), (768, 3072), (768, 2304), (1400, 768), (768, 1400), (1536, 768)):
    a = torch.randn(M, Kd, device=dev).bfloat16(); w = torch.randn(N, Kd, device=dev).bfloat16(); o = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    ms = t(lambda: K.gemm(a, w, out=o))
    print(f"NT M={M} N={N} K={Kd}: {ms * 1e3:.1f} us  {2.0 * M * N * Kd / ms / 1e9:.0f} TFLOP/s", flush=True)
