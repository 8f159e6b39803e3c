#!/usr/bin/env python3
"""bench.py -- headline benchmark of the ADT hot path on MI355X.

Contract (driver): ``python bench.py --gpus N --steps K --warmup W``; for N > 1 it
is launched under ``torch.distributed.run`` with one rank per GPU.  W untimed
steps, then exactly K timed steps bracketed by barrier + synchronize, MAX over
ranks, rank 0 prints ONE JSON line.

A "step" is one pass of the hot path over one batch of synthetic input that is
already resident in HBM.  Workloads (``--workload``):

  logmel  BASELINE config 2: fused STFT->log-mel, 256 clips x 10 s @ 16 kHz per GPU.

Every rank works on its own batch (weak scaling, no data-path collective).
The line also carries ``roofline`` (dominant kernel: algorithmic bytes per launch /
HIP-event time per launch, vs the 8 TB/s HBM peak) and, at N = 1, ``cpu_baseline``
(the oracle restatement timed on the host cores on a bounded sample).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP32_VECTOR_PEAK_TF = 157.3


def logmel_setup(dev, seed):
    from adt_str_amd.frontend import ComputeMelSpectrogram
    B, L, sr = 256, 160000, 16000
    g = torch.Generator().manual_seed(1234 + seed)
    wave = (torch.randn(B, L, generator=g) * 0.05)
    t = torch.arange(L) / sr
    for b in range(0, B, 4):                      # decaying bursts on a quarter of the clips
        t0 = float(torch.rand(1, generator=g)) * 9.0
        wave[b] += 0.5 * torch.exp(-(t - t0).clamp(min=0) * 30.0) * (t >= t0) * torch.sin(2 * torch.pi * 180.0 * (t - t0))
    wave[::16] = 0.0                              # every 16th clip silent
    wave = wave.clamp_(-1, 1)
    mod = ComputeMelSpectrogram(sr, 2048, 0.01, 128)
    wave_d = wave.to(dev)
    F = mod(wave_d[:1]).shape[1]
    algo_bytes = B * (4 * L + 4 * F * 128)        # read the wave once + write the output once
    # per frame: real FFT 2.5*N*log2(N) + power 3/bin + banded mel 4/bin + log/scale 3/mel
    algo_flops = B * F * (2.5 * 2048 * 11 + 3 * 1025 + 4 * 1025 + 3 * 128)
    return {"step": lambda: mod(wave_d), "units": B, "unit_name": "clips", "algo_bytes": algo_bytes,
            "algo_flops": algo_flops, "cpu_input": wave, "F": F,
            "config": {"workload": "logmel config[1]: 256 clips x 10 s @ 16 kHz -> [256,986,128], n_fft 2048, hop 160",
                       "clips_per_gpu": B, "samples": L, "sample_rate": sr}}


def logmel_cpu_baseline(wave, budget_s=12.0):
    from oracle import logmel as o_logmel
    torch.set_num_threads(min(os.cpu_count() or 1, 16))   # 16 = the reference's DataLoader worker count (setting-1.yaml:11)
    n = 64
    o_logmel.logmel(wave[:8], 16000, 2048, 0.01, 128)
    done, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        i = (done // n) % (wave.shape[0] // n)
        o_logmel.logmel(wave[i * n:(i + 1) * n], 16000, 2048, 0.01, 128)
        done += n
    dt = time.perf_counter() - t0
    return {"value": done / dt, "unit": "clips/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{done} clips of the same batch in {dt:.1f} s (oracle/logmel.py: torch.stft + dense mel matmul, fp32)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="logmel", choices=["logmel"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})")
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU path)"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)

    wl = logmel_setup(dev, seed=rank)
    step = wl["step"]

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        step()
    ev1.record()
    fence()
    dt = time.perf_counter() - t0
    kern_ms = ev0.elapsed_time(ev1) / args.steps          # HIP events on the stream the kernel runs on
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    if rank == 0:
        units = wl["units"] * world * args.steps
        achieved = wl["algo_bytes"] / (kern_ms * 1e-3) / 1e9
        line = {
            "metric": "ADT hot path clips/sec (10 s @16 kHz), log-mel front end stage",
            "value": units / dt, "unit": "clips/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic", "config": dict(wl["config"], parallelism=f"dp{world}"),
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "kernel": "adt::logmel_kernel", "kernel_ms": kern_ms,
                         "algorithmic_bytes_per_launch": wl["algo_bytes"],
                         "fp32_vector_tflops": wl["algo_flops"] / (kern_ms * 1e-3) / 1e12,
                         "fp32_vector_frac": wl["algo_flops"] / (kern_ms * 1e-3) / 1e12 / FP32_VECTOR_PEAK_TF},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = logmel_cpu_baseline(wl["cpu_input"])
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
