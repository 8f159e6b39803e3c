#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes: mean counter value per launch of the kernels whose name contains a pattern.
usage: pmc_summary.py <kernel-name-substring> <dir> [<dir> ...]  ->  JSON on stdout"""
import csv
import glob
import json
import sys

pat, dirs = sys.argv[1], sys.argv[2:]
out = {}
for d in dirs:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = {}
        for r in csv.DictReader(open(f)):
            if pat not in r["Kernel_Name"]:
                continue
            acc.setdefault(r["Counter_Name"], {}).setdefault(r["Dispatch_Id"], 0.0)
            acc[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
        for name, per in acc.items():
            vals = list(per.values())
            out[name] = {"launches": len(vals), "mean": sum(vals) / len(vals)}
print(json.dumps(out, indent=1))
