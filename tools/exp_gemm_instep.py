"""Experiment: the encoder layer's NT GEMMs timed inside the sequence the training step runs (LayerNorm -> QKV -> out-proj ->
LayerNorm -> FFN1 -> FFN2, then the backward data-gradient chain), per launch with HIP events.  Back-to-back loops of one launch
mislead: non-temporal output stores and a register-direct (v_permlane16_swap) epilogue won 5-13 % there and LOST here (round 2:
2.25-2.30 vs 2.13 ms for the eight launches), so epilogue changes are judged with this script.
    python tools/exp_gemm_instep.py; ADT_GEMM_GENERIC=1 python tools/exp_gemm_instep.py      (same gpurun call: boxes differ)"""
import os, sys, torch
if os.environ.get("ADT_GEMM_GENERIC", None) == "":
    del os.environ["ADT_GEMM_GENERIC"]
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adt_str_amd import kernels as K
dev = torch.device("cuda:0")
M = 63104
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s, sc=0.5: torch.randn(*s, device=dev, generator=g) * sc
x32 = rnd(M, 768); gamma, beta = torch.ones(768, device=dev), torch.zeros(768, device=dev)
wqkv, wo, w1, w2 = rnd(2304, 768, sc=0.03).bfloat16(), rnd(768, 768, sc=0.03).bfloat16(), rnd(3072, 768, sc=0.03).bfloat16(), rnd(768, 3072, sc=0.03).bfloat16()
w1t, w2t, wot, wqkvt = w1.t().contiguous(), w2.t().contiguous(), wo.t().contiguous(), wqkv.t().contiguous()
bqkv, b768, b3072 = rnd(2304), rnd(768), rnd(3072)
site = (0.1, 3)
names = ["qkv", "out-proj", "ffn1", "ffn2", "d(ffn2)xfactor+colsum", "d(ffn1)+res", "d(out-proj)", "d(qkv)+res"]
def seq(n, record):
    out = []
    for _ in range(n):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(16)]
        y32, y16 = K.layernorm_fwd(x32, gamma, beta)[:2]
        ev[0].record(); qkv = K.gemm(y16, wqkv, bias=bqkv); ev[1].record()
        a16 = qkv[:, :768].contiguous()                                    # stands in for the attention output
        ev[2].record(); x1 = K.gemm(a16, wo, out_dtype=torch.float32, bias=b768, residual=y32, drop=site); ev[3].record()
        z32, z16 = K.layernorm_fwd(x1, gamma, beta)[:2]
        u = torch.empty(M, 3072, device=dev, dtype=torch.bfloat16)
        ev[4].record(); h = K.gemm(z16, w1, bias=b3072, act=1, act_grad_out=u, drop=site); ev[5].record()
        ev[6].record(); x2 = K.gemm(h, w2, out_dtype=torch.float32, bias=b768, residual=z32, drop=site); ev[7].record()
        o32, o16 = K.layernorm_fwd(x2, gamma, beta)[:2]                    # o16 stands in for the incoming gradient (bf16 [M, 768])
        cs = torch.empty(3072, device=dev)
        ev[8].record(); dh = K.gemm(o16, w2t, act_grad=u, colsum_out=cs); ev[9].record()
        ev[10].record(); dz = K.gemm(dh, w1t, out_dtype=torch.float32, residual=o32); ev[11].record()
        d16 = K.layernorm_fwd(dz, gamma, beta)[1]
        ev[12].record(); da = K.gemm(d16, wot); ev[13].record()
        dqkv = torch.cat([da, da, da], dim=1)                              # stands in for the attention backward's packed dQ|dK|dV
        ev[14].record(); dx = K.gemm(dqkv, wqkvt, out_dtype=torch.float32, residual=dz); ev[15].record()
        if record: out.append(ev)
    torch.cuda.synchronize()
    return out
seq(6, False)
evs = seq(14, True)
tot = 0.0
line = []
for i, nme in enumerate(names):
    ts = sorted(e[2 * i].elapsed_time(e[2 * i + 1]) for e in evs)
    med = ts[len(ts) // 2]; tot += med
    line.append(f"{nme} {med:.3f}")
print(f"{'generic kernel' if os.environ.get('ADT_GEMM_GENERIC') else 'specialised forms'}: total {tot:.3f} ms | " + ", ".join(line), flush=True)
