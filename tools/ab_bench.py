#!/usr/bin/env python3
"""A/B kernels in ONE gpurun call on ONE device: run bench.py alternately with different builds of the library
(ISSCABAC_LIB=...) and print the per-kernel medians.  usage: python tools/ab_bench.py libA.so libB.so ... [--rounds 3]"""
import json
import os
import statistics
import subprocess
import sys

workload = "config3"
if "--workload" in sys.argv:
    workload = sys.argv[sys.argv.index("--workload") + 1]
    sys.argv.remove("--workload"); sys.argv.remove(workload)
extra = []
if "--" in sys.argv:  # everything behind `--` goes to bench.py (e.g. -- --streams 65536 --tuning latency)
    i = sys.argv.index("--")
    extra = sys.argv[i + 1:]
    del sys.argv[i:]
libs = [a for a in sys.argv[1:] if not a.startswith("--")]
rounds = 3
if "--rounds" in sys.argv:
    rounds = int(sys.argv[sys.argv.index("--rounds") + 1])
    libs = [a for a in libs if a != str(rounds)]
res = {l: {"encode": [], "decode": [], "gather": [], "value": []} for l in libs}
for r in range(rounds):
    for l in libs:
        env = dict(os.environ, ISSCABAC_LIB=os.path.abspath(l), ISSCABAC_BENCH_FULL_CAP="1")
        out = subprocess.run([sys.executable, "bench.py", "--steps", "4", "--warmup", "1", "--cpu-seconds", "0", "--no-verify", "--workload", workload] + extra,
                             env=env, capture_output=True, text=True)
        try:
            d = json.loads(out.stdout.strip().splitlines()[-1])
        except Exception:
            print(l, "FAILED", out.stderr[-500:])
            continue
        res[l]["encode"].append(d["kernels_ms"]["encode"])
        res[l]["decode"].append(d["kernels_ms"]["decode"])
        res[l]["gather"].append(d["kernels_ms"]["gather"] + d["kernels_ms"]["scan"])
        res[l]["value"].append(d["value"])
for l in libs:
    x = res[l]
    if x["value"]:
        print("%-40s enc %.3f ms  dec %.3f ms  scan+gather %.3f ms  value %.1f Gbin/s   (medians of %d)" % (
            os.path.basename(l), statistics.median(x["encode"]), statistics.median(x["decode"]), statistics.median(x["gather"]),
            statistics.median(x["value"]), len(x["value"])))
