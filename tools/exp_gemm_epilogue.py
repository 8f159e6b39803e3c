"""A/B of the persistent NT GEMM's epilogue paths on the forms the training step launches (M = 63104).

Run once per setting inside ONE gpurun call (boxes differ by up to 10 %):
    for e in "" 0 2 3; do ADT_GEMM_EPI=$e python tools/exp_gemm_epilogue.py; done
ADT_GEMM_EPI unset/empty: the product's choice (specialised kernels of ADT_NT256_FORMS, gemm.hip); 0 / 2 / 3: the generic kernel
forced onto the LDS-transposition / direct + non-temporal / direct path."""
import os, sys, torch
if os.environ.get("ADT_GEMM_EPI", None) == "":
    del os.environ["ADT_GEMM_EPI"]
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adt_str_amd import kernels as K


def timeit(fn, n=60, warm=30):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


dev = torch.device("cuda:0")
M = 63104
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s, sc=0.5: torch.randn(*s, device=dev, generator=g) * sc
a768, a3072 = rnd(M, 768).bfloat16(), rnd(M, 3072).bfloat16()
w1, w2, wo, wqkv = rnd(3072, 768, sc=0.03).bfloat16(), rnd(768, 3072, sc=0.03).bfloat16(), rnd(768, 768, sc=0.03).bfloat16(), rnd(2304, 768, sc=0.03).bfloat16()
b3072, b768, b2304 = rnd(3072), rnd(768), rnd(2304)
u, o3072 = (torch.empty(M, 3072, device=dev, dtype=torch.bfloat16) for _ in range(2))
o2304 = torch.empty(M, 2304, device=dev, dtype=torch.bfloat16)
o768f, res, cs = torch.empty(M, 768, device=dev), rnd(M, 768), torch.empty(3072, device=dev)
site = (0.1, 3)
cases = {
    "plain 3072x768": (lambda: K.gemm(a768, w1, out=o3072), 3072 * 768),
    "qkv + bias": (lambda: K.gemm(a768, wqkv, out=o2304, bias=b2304), 2304 * 768),
    "ffn1 + bias + gelu + dropout + saved factor": (lambda: K.gemm(a768, w1, out=o3072, bias=b3072, act=1, act_grad_out=u, drop=site), 3072 * 768),
    "ffn2 + bias + dropout + residual (fp32 out)": (lambda: K.gemm(a3072, w2, out=o768f, out_dtype=torch.float32, bias=b768, residual=res, drop=site), 3072 * 768),
    "out-proj + bias + dropout + residual": (lambda: K.gemm(a768, wo, out=o768f, out_dtype=torch.float32, bias=b768, residual=res, drop=site), 768 * 768),
    "dx K=3072 + residual (fp32 out)": (lambda: K.gemm(a3072, w2, out=o768f, out_dtype=torch.float32, residual=res), 3072 * 768),
    "ffn dgrad x saved factor": (lambda: K.gemm(a768, w1, out=o3072, act_grad=u), 3072 * 768),
    "ffn dgrad x saved factor + colsum": (lambda: K.gemm(a768, w1, out=o3072, act_grad=u, colsum_out=cs), 3072 * 768),
}
print(f"ADT_GEMM_EPI={os.environ.get('ADT_GEMM_EPI', '(product)')}")
for name, (fn, nk) in cases.items():
    t = min(timeit(fn), timeit(fn))
    print(f"  {name:46s} {t:.4f} ms  {2.0 * M * nk / t / 1e9:6.0f} TFLOP/s", flush=True)
