"""Demonstration: a kernel that holds a few CUs (the stand-in for an RCCL collective, adt_debug_occupy) while the persistent GEMMs
run on another stream.  The GEMMs take their tiles from work counters: they neither wait for the occupier nor lose more than
the CUs it holds (measured: 0.338 ms alone, 0.337-0.341 ms beside 8-16 occupied CUs)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adt_str_amd import kernels as K, _ffi
dev = torch.device("cuda:0")
M = 63104
g = torch.Generator(device=dev).manual_seed(0)
a = (torch.randn(M, 768, device=dev, generator=g) * 0.5).bfloat16()
w = (torch.randn(3072, 768, device=dev, generator=g) * 0.03).bfloat16()
o = torch.empty(M, 3072, device=dev, dtype=torch.bfloat16)
side = torch.cuda.Stream()
def run(n_occ, lds, micros=20000, n=20):
    for _ in range(5): K.gemm(a, w, out=o)
    torch.cuda.synchronize()
    if n_occ: _ffi.call("adt_debug_occupy", n_occ, lds, micros, side.cuda_stream)
    time.sleep(0.002)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): K.gemm(a, w, out=o)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
print(f"no occupier: {run(0, 0):.3f} ms per GEMM")
for n_occ, lds in ((8, 16384), (8, 65536), (16, 16384), (32, 65536)):
    print(f"{n_occ} occupier workgroups x {lds // 1024} KB LDS for 20 ms: {run(n_occ, lds):.3f} ms per GEMM")
