#!/usr/bin/env python3
"""Every launch of ONE training step in order, from a rocprofv3 kernel trace:
    rocprofv3 --kernel-trace --output-format csv -d <dir> -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-e2e --no-clap
    python tools/step_timeline.py <dir> [step [delimiter]]
Steps are delimited by adamw_kernel launches (the roofline loops of bench.py run after the last one and are not part of any step);
another delimiter kernel can be named (l2_normalize_kernel ends a forward of the CLAP tower: tools/prof_clap_forward.py).
Per launch: start offset inside the step, duration, idle gap in front, workgroups, name."""
import csv
import glob
import re
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], (int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])) // max(1, int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"]))) for r in csv.DictReader(open(f))),
              key=lambda r: r[0])
delim = sys.argv[3] if len(sys.argv) > 3 else "adamw_kernel"
ends = [i for i, r in enumerate(rows) if delim in r[2]]
k = int(sys.argv[2]) if len(sys.argv) > 2 and int(sys.argv[2]) >= 0 else len(ends) - 2
seg = rows[ends[k] + 1:ends[k + 1] + 1]
t0, prev = seg[0][0], seg[0][0]
short = lambda n: re.sub(r"\(.*", "", n.replace("void ", "").replace("adt::", ""))[:70]
agg = {}
for s, e, n, wg in seg:
    print(f"{(s - t0) / 1e3:9.1f} us  {(e - s) / 1e3:7.1f} us  gap {max(0, s - prev) / 1e3:5.1f}  wgs {wg:6d}  {short(n)}")
    prev = e
    a = agg.setdefault(short(n), [0, 0.0])
    a[0] += 1
    a[1] += (e - s) / 1e3
print(f"wall {(seg[-1][1] - t0) / 1e6:.3f} ms, busy {sum(e - s for s, e, _, _ in seg) / 1e6:.3f} ms, {len(seg)} launches")
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"  {t / 1e3:7.3f} ms  {c:4d} x {t / c:7.1f} us  {n}")
