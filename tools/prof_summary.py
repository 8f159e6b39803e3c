#!/usr/bin/env python3
"""Per-kernel totals of a rocprofv3 --kernel-trace run stored as a rocpd SQLite file: usage prof_summary.py <results.db> [n_iterations]."""
import collections
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
n = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
rows = cur.execute(f"select s.display_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id = s.id").fetchall()
agg = collections.OrderedDict()
for name, st, en in rows:
    a = agg.setdefault(name[:120], [0, 0])
    a[0] += 1
    a[1] += en - st
tot = sum(v[1] for v in agg.values())
print(f"{'ms/iter':>9} {'calls/iter':>10} {'share':>6}  kernel")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:30]:
    print(f"{v[1] / n / 1e6:9.3f} {v[0] / n:10.1f} {100 * v[1] / tot:5.1f}%  {k}")
print(f"{tot / n / 1e6:9.3f} total")
