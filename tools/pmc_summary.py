#!/usr/bin/env python3
"""Summarise the rocprofv3 --pmc passes written by tools/pmc_profile.sh: per kernel, counter averages per launch.
usage: python tools/pmc_summary.py gpurun_out/pmc_r01 [out.json]"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def main():
    root = sys.argv[1]
    acc = defaultdict(lambda: defaultdict(list))
    for path in glob.glob(os.path.join(root, "*", "*", "*_counter_collection.csv")):
        per_dispatch = defaultdict(float)
        names = {}
        with open(path) as f:
            for r in csv.DictReader(f):
                key = (r["Dispatch_Id"], r["Counter_Name"])
                per_dispatch[key] += float(r["Counter_Value"])
                names[r["Dispatch_Id"]] = r["Kernel_Name"]
        for (d, c), v in per_dispatch.items():
            k = names[d]
            short = "k_" + k.split("k_")[1].split("(")[0] if "k_" in k else k[:40]
            acc[short][c].append(v)
    out = {}
    for k in sorted(acc):
        out[k] = {c: sum(v) / len(v) for c, v in sorted(acc[k].items())}
        out[k]["launches_seen"] = max(len(v) for v in acc[k].values())
    txt = json.dumps(out, indent=1)
    print(txt)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(txt)


if __name__ == "__main__":
    main()
