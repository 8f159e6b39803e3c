"""Experiment: per-tile timeline of the persistent NT-256 kernel (s_memtime stamps of every wave's lane 0)."""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adt_str_amd import kernels as K
dev = torch.device("cuda:0")
M = 63104
g = torch.Generator(device=dev).manual_seed(0)
a = (torch.randn(M, 768, device=dev, generator=g) * 0.5).bfloat16()
w = (torch.randn(3072, 768, device=dev, generator=g) * 0.03).bfloat16()
bias = torch.randn(3072, device=dev, generator=g)
o = torch.empty(M, 3072, device=dev, dtype=torch.bfloat16); u = torch.empty_like(o)
tr = torch.zeros(256 * 8 * 16 * 8, device=dev, dtype=torch.int64)
names = ["K loop", "drain", "handoff+prologue", "epilogue", "final vmcnt(0)", "final barrier"]
for dbg in (0, 1):
  os.environ["ADT_GEMM_DBG"] = str(dbg)
  for label, fn in (("plain", lambda: K.gemm(a, w, out=o)), ("ffn1+gelu+drop+factor", lambda: K.gemm(a, w, out=o, bias=bias, act=1, act_grad_out=u, drop=(0.1, 3)))):
    os.environ.pop("ADT_GEMM_TRACE_PTR", None)
    for _ in range(20): fn()
    torch.cuda.synchronize()
    os.environ["ADT_GEMM_TRACE_PTR"] = str(tr.data_ptr())
    tr.zero_(); fn(); torch.cuda.synchronize()
    os.environ.pop("ADT_GEMM_TRACE_PTR", None)
    t = tr.cpu().numpy().reshape(256, 8, 16, 8).astype(np.float64)
    d = np.diff(t[:, :, 2:9, :7], axis=3)          # [wg, wave, tile, 6 intervals]
    tot = t[:, 0, 3:9, 0] - t[:, 0, 2:8, 0]
    print(f"dbg={dbg} {label}: per tile {tot.mean():8.0f} ticks", flush=True)
    for wv in range(8):
        print(f"   wave {wv}: " + ", ".join(f"{n} {d[:, wv, :, i].mean():7.0f}" for i, n in enumerate(names)))
    # spread of tile starts across workgroups at tile 5
    st = t[:, 0, 5, 0]; st = st[st > 0]
    print(f"   tile-5 start spread over workgroups: std {st.std():.0f} ticks, range {st.max() - st.min():.0f}")
