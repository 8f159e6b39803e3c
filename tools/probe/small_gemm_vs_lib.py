import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from adt_str_amd import kernels as K
dev = "cuda:0"
def timeit(fn, n=50):
    for _ in range(3): fn()
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side): fn()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (5 * n) * 1e3
for M, N, Kd in [(8192, 768, 768), (8192, 768, 3072), (8192, 768, 1400), (8192, 2304, 768), (32768, 768, 768), (32768, 768, 3072), (131072, 384, 1536)]:
    a = torch.randn(M, Kd, device=dev).bfloat16(); w = torch.randn(N, Kd, device=dev).bfloat16()
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    t1 = timeit(lambda: K.gemm(a, w, out=out)); t2 = timeit(lambda: torch.matmul(a, w.t(), out=out))
    fl = 2.0 * M * N * Kd / 1e6
    print(f"NT M={M} N={N} K={Kd}: this repo {t1:.1f} us ({fl / t1:.0f} TF/s) | torch.matmul {t2:.1f} us ({fl / t2:.0f} TF/s)", flush=True)
