#!/usr/bin/env python3
"""Copy one evidence run (bench.json, rocprof kernel stats, PMC summary/table made on the GPU box under gpurun_out/<dir>)
into profiles/<round>/final_* and update profiles/pmc_traffic.json.  usage: python tools/refresh_profiles.py gpurun_out/r01d r01"""
import glob, json, shutil, sys
src, rnd = sys.argv[1], sys.argv[2]
dst = "profiles/%s/" % rnd
shutil.copy(src + "/bench.json", dst + "final_bench.json")
shutil.copy(glob.glob(src + "/stats/*/*_kernel_stats.csv")[0], dst + "final_kernel_stats.csv")
shutil.copy(src + "/pmc_summary.json", dst + "final_pmc_summary.json")
shutil.copy(src + "/pmc_table.txt", dst + "final_pmc_table.txt")
d = json.load(open(src + "/pmc_summary.json"))
t = json.load(open("profiles/pmc_traffic.json"))
for k, x in d.items():
    for name in ("k_decode", "k_encode"):
        if k.startswith(name + "<"):
            f, w = x["FETCH_SIZE"] * 1024, x["WRITE_SIZE"] * 1024
            t["config3"][name].update(fetch_bytes=f, write_bytes=w, hbm_bytes_per_launch=f + w)
json.dump(t, open("profiles/pmc_traffic.json", "w"), indent=1)
b = json.load(open(src + "/bench.json"))
print("value %.1f Gbin/s" % b["value"], b["kernels_ms"], "frac %.3f" % b["roofline"]["frac"], "cpu %.4f" % b["cpu_baseline"]["value"])
print(open(dst + "final_kernel_stats.csv").read()[:520])
