import sys, time, random
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from adt_str_amd.bank import OneShotBank, synthetic_tree
from adt_str_amd.synth import SynthDrum, SynthDrumConfig
sr=16000
bank = OneShotBank.from_tree(synthetic_tree(7, sr), sr)
def mk(p, probs=(0.5, 0.5, 0.5)):
    return SynthDrum(SynthDrumConfig(input_sec=10.0, time_res=0.01, win_length=2048, sample_rate=sr, oneshot_path="synthetic", similarity_threshold=0.8,
        max_hat_std_velocity=0.15, max_hat_mean_velocity=0.1, max_cymbals_std_velocity=0.15, max_cymbals_mean_velocity=0.65, ADTOF_mapping=False,
        mixup_range=0.8, use_fx_prob=p, use_reverb_prob=probs[0], use_limiter_prob=probs[2], use_compression_prob=probs[1]), bank=bank, device="cuda:0")
rng=np.random.default_rng(0); random.seed(0); torch.manual_seed(0)
notes=bench.synthetic_notes(rng, 64)
for p, probs in ((0.0, (0.5, 0.5, 0.5)), (0.3, (0.5, 0.5, 0.5)), (1.0, (1, 0, 0)), (1.0, (0, 1, 0)), (1.0, (0, 0, 1)), (1.0, (1, 1, 1))):
    sd=mk(p, probs); plan=sd.plan(notes)
    out=torch.empty((64,160000),device="cuda:0")
    for _ in range(3): sd.render_plan(plan, width=160000, out=out)
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(10): sd.render_plan(plan, width=160000, out=out)
    torch.cuda.synchronize()
    # overlap check: a long matmul on the main stream, then a render without a caller buffer (own stream) -> the two should overlap
    a=torch.randn(8192,8192,device="cuda:0",dtype=torch.bfloat16)
    def busy():
        for _ in range(12): torch.mm(a,a)
    busy(); torch.cuda.synchronize(); t1=time.perf_counter(); busy(); torch.cuda.synchronize(); tb=time.perf_counter()-t1
    t1=time.perf_counter(); busy(); r=sd.render_plan(plan, width=160000); torch.cuda.synchronize(); tboth=time.perf_counter()-t1
    print(f"   main-stream work alone {tb*1e3:.2f} ms; + render on the synth stream {tboth*1e3:.2f} ms")
    nfx=0 if plan.fx is None else int((plan.fx["flags"]!=0).sum())
    print(f"use_fx_prob={p} (reverb, comp, limiter probs {probs}): {nfx} FX clips of 64, render {(time.perf_counter()-t0)*100:.2f} ms/batch")
