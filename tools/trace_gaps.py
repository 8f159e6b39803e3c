#!/usr/bin/env python3
"""Idle time between kernels of the training step, from a rocprofv3 kernel trace:
    rocprofv3 --kernel-trace --output-format csv -d <dir> -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-e2e --no-clap
    python tools/trace_gaps.py <dir>
Steps are delimited by adamw_kernel launches; per step: wall time from the first kernel's start to the last kernel's end, the sum of
kernel durations, the idle remainder and how it is distributed over gap sizes."""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))), key=lambda r: r[0])
ends = [i for i, r in enumerate(rows) if "adamw_kernel" in r[2]]
for a, b in zip(ends[:-1], ends[1:]):
    seg = rows[a + 1:b + 1]
    wall = (seg[-1][1] - seg[0][0]) / 1e6
    busy = sum(e - s for s, e, _ in seg) / 1e6
    gaps = [max(0, seg[i + 1][0] - seg[i][1]) / 1e3 for i in range(len(seg) - 1)]
    big = sorted(((g, seg[i][2][:50], seg[i + 1][2][:50]) for i, g in enumerate(gaps)), reverse=True)[:6]
    print(f"step: {len(seg)} kernels, wall {wall:.2f} ms, busy {busy:.2f} ms, idle {wall - busy:.2f} ms; gaps: median {sorted(gaps)[len(gaps) // 2]:.1f} us, "
          f">10us: {sum(g > 10 for g in gaps)}, >50us: {sum(g > 50 for g in gaps)}")
    for g, x, y in big[:4]:
        print(f"    {g:8.1f} us between {x} -> {y}")
