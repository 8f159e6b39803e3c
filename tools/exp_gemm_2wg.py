#!/usr/bin/env python3
"""Experiment (round 6): the persistent NT GEMM as 256 x 128 tiles with TWO workgroups per CU (gemm_nt_2wg_kernel) against the 256 x 256
one-workgroup form (gemm_nt_256_kernel), in ONE process on one box: ADT_GEMM_ENV_DYNAMIC=1 makes the library re-read ADT_GEMM_NT on every
call, so the two kernels alternate launch loop by launch loop.

  1. agreement: both kernels add the same 32-deep MFMA partial sums in the same order, so every epilogue form must agree BIT FOR BIT;
  2. timings of the training step's NT forms at the encoder shapes (M = 64 x 986), the decoder / CLAP mid-size shapes, and the
     stagger sweep of the 2-WG form (ADT_GEMM_STAGGER2 is read once per process: pass --stagger N to set it for this run).

usage: python tools/exp_gemm_2wg.py [--stagger TICKS] [--quick]"""
import argparse
import os
import sys

ap = argparse.ArgumentParser()
ap.add_argument("--stagger", type=int, default=None)
ap.add_argument("--quick", action="store_true")
ap.add_argument("--mid", action="store_true", help="only the mid-size shapes, with the library's own tile choice as the third column")
args = ap.parse_args()
os.environ["ADT_GEMM_ENV_DYNAMIC"] = "1"
if not args.mid:
    os.environ["ADT_GEMM_TILE"] = "256"          # (read once per process) the persistent kernels for every shape, small M included
if args.stagger is not None:
    os.environ["ADT_GEMM_STAGGER2"] = str(args.stagger)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from adt_str_amd import kernels as K  # noqa: E402

dev = "cuda:0"


def use(form):
    os.environ["ADT_GEMM_NT"] = form


def timeit(fn, n=30, warm=15):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def forms_for(M, d=768, ffn=3072):
    g = torch.Generator(device=dev).manual_seed(M)
    rn = lambda *s: torch.randn(s, device=dev, generator=g)
    a = rn(M, d).bfloat16()
    a2 = rn(M, ffn).bfloat16()
    w1 = (rn(ffn, d) * 0.05).bfloat16()
    w2 = (rn(d, ffn) * 0.05).bfloat16()
    wq = (rn(3 * d, d) * 0.05).bfloat16()
    wo = (rn(d, d) * 0.05).bfloat16()
    b1, bo, bq = rn(ffn), rn(d), rn(3 * d)
    res = rn(M, d)
    fac = rn(M, ffn).bfloat16()
    site = K.drop_site(0.1, 1, 5)
    outs = {}

    def ffn1():
        u = torch.empty((M, ffn), dtype=torch.bfloat16, device=dev)
        h = K.gemm(a, w1, bias=b1, act=1, act_grad_out=u, drop=site)
        return h, u

    def dgrad_factor():
        cs = torch.empty(ffn, device=dev)
        y = K.gemm(a, w1, act_grad=fac, colsum_out=cs)
        return y, cs

    forms = {
        "FFN-1 (bias+GELU+dropout+saved factor)": ffn1,
        "bare N=3072 K=768": lambda: K.gemm(a, w1),
        "bare N=768 K=3072": lambda: K.gemm(a2, w2),
        "QKV (bias) N=2304": lambda: K.gemm(a, wq, bias=bq),
        "out-proj (bias+dropout+residual, fp32)": lambda: K.gemm(a, wo, bias=bo, out_dtype=torch.float32, drop=site, residual=res),
        "FFN-2 (bias+dropout+residual, fp32)": lambda: K.gemm(a2, w2, bias=bo, out_dtype=torch.float32, drop=site, residual=res),
        "FFN dgrad x saved factor + colsum": dgrad_factor,
        "dx K=3072 + residual fp32": lambda: K.gemm(a2, w2, out_dtype=torch.float32, residual=res),
    }
    return forms, outs


def flat(o):
    return o if isinstance(o, tuple) else (o,)


def main():
    import inspect
    sig = inspect.signature(K.gemm).parameters
    print("K.gemm parameters:", ", ".join(sig), flush=True)
    if args.mid:
        return mid()
    # ---- 1. agreement
    bad = 0
    for M in ((1000, 4360, 63104) if not args.quick else (4360,)):
        forms, _ = forms_for(M)
        for name, fn in forms.items():
            use("256")
            r256 = [t.clone() for t in flat(fn())]
            for other in ("2wg", "ring"):
                use(other)
                r2 = [t.clone() for t in flat(fn())]
                same = all(torch.equal(x, y) for x, y in zip(r256, r2))
                worst = max(float((x.float() - y.float()).abs().max()) for x, y in zip(r256, r2))
                fin = all(bool(torch.isfinite(y.float()).all()) for y in r2)
                print(f"agree M={M:6d} {name:42s} {other:4s} bitwise={same} max|diff|={worst:.3e} finite={fin}", flush=True)
                bad += 0 if same else 1
                del r2
            del r256
        del forms
        torch.cuda.empty_cache()
    print("AGREEMENT:", "all forms bit for bit" if bad == 0 else f"{bad} form(s) differ", flush=True)
    # ---- 2. timings, alternating
    forms, _ = forms_for(63104)
    for rep in range(2):
        for name, fn in forms.items():
            t = {}
            for _ in range(2):
                for f in ("256", "2wg", "ring"):
                    use(f)
                    t.setdefault(f, []).append(timeit(fn))
            b = {f: min(x) for f, x in t.items()}
            print(f"time rep{rep} {name:42s} phases-256: {t['256'][0]:.4f} {t['256'][1]:.4f}   2wg: {t['2wg'][0]:.4f} {t['2wg'][1]:.4f}   ring-256: {t['ring'][0]:.4f} {t['ring'][1]:.4f} ms"
                  f"   2wg/256 {b['2wg'] / b['256']:.3f}  ring/256 {b['ring'] / b['256']:.3f}", flush=True)


def mid():
    # mid-size shapes (decoder M = 8192, CLAP M = 32768 / 131072): bare products
    g = torch.Generator(device=dev).manual_seed(1)
    for (M, N, Kd) in ((8192, 2304, 768), (8192, 3072, 768), (8192, 1400, 768), (8192, 768, 3072), (32768, 768, 3072), (32768, 3072, 768), (32768, 2304, 768),
                       (131072, 384, 1536), (131072, 1536, 384), (8192, 8192, 8192)):
        a = torch.randn((M, Kd), device=dev, generator=g).bfloat16()
        w = torch.randn((N, Kd), device=dev, generator=g).bfloat16()
        o = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
        os.environ.pop("ADT_GEMM_NT", None); t0 = timeit(lambda: K.gemm(a, w, out=o))
        use("256"); t1 = timeit(lambda: K.gemm(a, w, out=o))
        use("2wg"); t2 = timeit(lambda: K.gemm(a, w, out=o))
        use("ring"); t3 = timeit(lambda: K.gemm(a, w, out=o))
        print(f"mid {M}x{N}x{Kd}: library's choice {t0 * 1e3:.1f} us   ADT_GEMM_NT=256 {t1 * 1e3:.1f}   2wg {t2 * 1e3:.1f}   ring {t3 * 1e3:.1f} us (these differ only where the shape takes a persistent kernel)", flush=True)
        del a, w, o


if __name__ == "__main__":
    main()
