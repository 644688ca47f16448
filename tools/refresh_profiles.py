#!/usr/bin/env python3
"""Copy one evidence run (tools/evidence_run.sh, made on the GPU box under gpurun_out/<dir>) into profiles/<round>/final_*
and update profiles/pmc_traffic.json.  usage: python tools/refresh_profiles.py gpurun_out/r02h r02"""
import glob, json, os, shutil, sys


def update_traffic(summary, note):
    """FETCH_SIZE / WRITE_SIZE of the coder kernels (KiB per launch) -> profiles/pmc_traffic.json, which bench.py quotes"""
    d = json.load(open(summary))
    t = json.load(open("profiles/pmc_traffic.json"))
    for name in ("k_decode", "k_encode"):
        # config 3 under automatic tuning runs ONE instance of each kernel: <= 64 contexts, throughput, not bucketed.  A summary
        # that holds others as well (small batches, > 128 contexts, bucketed launches) must not overwrite it silently.
        want = name + "<1, false, false>"
        hits = [k for k in d if k.startswith(name + "<") and "FETCH_SIZE" in d[k]]
        if want not in hits:
            if hits:
                raise SystemExit("refresh_profiles: %s not in the PMC summary (found %s)" % (want, hits))
            continue
        x = d[want]
        # FETCH_SIZE tallies 64 bytes per L2 line fill although a fill moves the whole 128-byte line whatever the request
        # asked for (profiles/r04/fetch_calibration.txt): reads = 2 x the counter.  WRITE_SIZE is exact.
        f, w = 2.0 * x["FETCH_SIZE"] * 1024, x["WRITE_SIZE"] * 1024
        t.setdefault("config3", {}).setdefault(name, {}).update(fetch_counter_bytes=x["FETCH_SIZE"] * 1024, fetch_bytes=f, write_bytes=w,
                                                                 hbm_bytes_per_launch=f + w, source=note)
    json.dump(t, open("profiles/pmc_traffic.json", "w"), indent=1)


if sys.argv[1] == "--traffic-only":  # on the GPU box, between the PMC passes and the bench line (tools/evidence_run.sh)
    update_traffic(sys.argv[2], "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this build, same evidence run; reads = 2 x FETCH_SIZE "
                   "(64 B tallied per 128-byte line fill: profiles/r04/fetch_calibration.txt).  This is L2 <-> fabric traffic: fills served by "
                   "the Infinity Cache are in it, and the HBM's own share cannot be measured with this pool's counters (profiles/r05/hbm_side.txt)")
    sys.exit(0)
src, rnd = sys.argv[1], sys.argv[2]
dst = "profiles/%s/" % rnd
os.makedirs(dst, exist_ok=True)
shutil.copy(src + "/bench.json", dst + "final_bench.json")
shutil.copy(glob.glob(src + "/stats/*/*_kernel_stats.csv")[0], dst + "final_kernel_stats.csv")
for f in ("pmc_summary.json", "pmc_table.txt", "other_shapes.txt", "s_sweep.txt", "s_sweep.json"):
    if os.path.exists(src + "/" + f):
        shutil.copy(src + "/" + f, dst + "final_" + f)
note = ("profiles/%s/final_pmc_summary.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; reads = 2 x FETCH_SIZE: the "
        "counter tallies 64 B per 128-byte line fill, calibrated on known byte counts in profiles/r04/fetch_calibration.txt).  L2 <-> fabric "
        "traffic: fills served by the Infinity Cache are in it; the HBM's own share is not measurable here (profiles/r05/hbm_side.txt)" % rnd)
update_traffic(src + "/pmc_summary.json", note)
b = json.load(open(src + "/bench.json"))
print("value %.1f Gbin/s" % b["value"], b["kernels_ms"], "frac %.3f" % b["roofline"]["frac"], "cpu %.4f" % b["cpu_baseline"]["value"])
print(open(dst + "final_kernel_stats.csv").read()[:700])
