import math, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from adt_str_amd import _ffi
from adt_str_amd.clap_encoder import pack_attn_block_weights, pack_rowblock_weights, rowblock, window_bias_layout
DEV = "cuda:0"
g = torch.Generator().manual_seed(7)
B, R, C, nh = 2, 16, 96, 4
M = B * R * R
x = (torch.randn((M, C), generator=g) * 1.2).to(DEV)
gamma, beta = torch.ones(C, device=DEV), torch.zeros(C, device=DEV)
scale = 1.0 / math.sqrt(24.0)
def run(wqkv, bqkv, wo, bo, bias, tag):
    biasl = window_bias_layout(bias)
    xr = x.clone()
    qkv = torch.empty((M, 3 * C), dtype=torch.bfloat16, device=DEV)
    rowblock(0, xr, pack_rowblock_weights(0, wqkv).to(DEV), 3 * C // 32, bqkv, ln=(gamma, beta), out16=qkv)
    ctx = torch.empty((M, C), dtype=torch.bfloat16, device=DEV)
    _ffi.call("adt_window_attn_fwd", qkv.data_ptr(), qkv.stride(0), ctx.data_ptr(), C, biasl.data_ptr(), 1, B, R, C, nh, 0, scale, 0)
    rowblock(1, xr, pack_rowblock_weights(1, wo).to(DEV), C // 32, bo, a16=ctx)
    xf = x.clone()
    wpk, qkvb = pack_attn_block_weights(wqkv, bqkv, wo, nh)
    _ffi.call("adt_htsat_attn_block", xf.data_ptr(), B, R, C, nh, 0, gamma.data_ptr(), beta.data_ptr(), 1e-5, wpk.data_ptr(), qkvb.data_ptr(),
              bo.data_ptr(), biasl.data_ptr(), 1, scale, 0)
    torch.cuda.synchronize()
    ur, uf = xr - x, xf - x
    d = (uf - ur)
    bad = ~torch.isfinite(uf)
    print(f"{tag}: ref |upd| max {ur.abs().max():.4f}; fused max {uf[~bad].abs().max() if (~bad).any() else float('nan'):.4g}; nonfinite {int(bad.sum())}; "
          f"max diff {d[~bad].abs().max():.4g}; rows with diff>0.05: {int((d.abs().amax(1) > 0.05).sum())}/{M}; cols: {(d.abs().amax(0) > 0.05).nonzero().flatten().tolist()[:12]}")
    return ur, uf
Z = lambda *s: torch.zeros(s, device=DEV)
Rn = lambda *s: (torch.randn(s, generator=g) ).to(DEV)
wqkv, bqkv, wo, bo = Rn(3 * C, C) / C ** 0.5, 0.2 * Rn(3 * C), Rn(C, C) / C ** 0.5, 0.1 * Rn(C)
bias = 0.5 * Rn(nh, 64, 64)
run(wqkv, bqkv, Z(C, C), bo, bias, "A wo=0")
w3 = wqkv.clone(); w3[2 * C:] = 0; b3 = bqkv.clone(); b3[2 * C:] = 1.0
run(w3, b3, wo, bo, bias, "F v == 1 (wv = 0, bv = 1)")
w4 = Z(3 * C, C); b4 = Z(3 * C); b4[2 * C:] = torch.arange(C, device=DEV).float() / 10
ur, uf = run(w4, b4, torch.eye(C, device=DEV), Z(C), Z(nh, 64, 64), "G q=k=0, v[d] = d/10, wo = I: update[c] = c/10")
torch.set_printoptions(linewidth=200, precision=4)
print("o of head 1 (expect 2.4 + d/10, zeros from d = 24):"); print(uf[0, :32]); print(uf[37, :32]); print(uf[40, :32])
print("P (st[0], st[1]) of token 0, 37 (expect 1 each):"); print(uf[0, 32:96]); print(uf[37, 32:96])
w2 = wqkv.clone(); w2[:2 * C] = 0; b2 = bqkv.clone(); b2[:2 * C] = 0
run(w2, b2, wo, bo, Z(nh, 64, 64), "B q=k=0, no bias (uniform attention)")
run(w2, b2, wo, bo, bias, "C q=k=0, with rel bias")
run(wqkv, bqkv, wo, bo, Z(nh, 64, 64), "D full, no rel bias")
ur, uf = run(wqkv, bqkv, wo, bo, bias, "E full")
print(ur[:2, :8]); print(uf[:2, :8])
