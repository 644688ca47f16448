#!/bin/bash
# round 6, run b: whole-line encoder tiles (TileLine4) -- parity, same-box A/B, FETCH / WRITE and clock per variant
O=gpurun_out/r06b; mkdir -p $O
ISSCABAC_LIB=$PWD/tools/ab/variant_line.so timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "throughput" > $O/parity_line.txt 2>&1; tail -3 $O/parity_line.txt
ISSCABAC_LIB=$PWD/tools/ab/variant_lineNT.so timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "throughput and (config3 or golden or ragged)" > $O/parity_lineNT.txt 2>&1; tail -3 $O/parity_lineNT.txt
python tools/ab_bench.py tools/ab/variant_base.so tools/ab/variant_line.so tools/ab/variant_lineB.so tools/ab/variant_lineC.so tools/ab/variant_lineNT.so --rounds 5 > $O/ab_line.txt 2>&1; cat $O/ab_line.txt
bash tools/pmc_fetch.sh $O/pmc tools/ab/variant_base.so tools/ab/variant_line.so tools/ab/variant_lineNT.so > $O/fetch.txt 2>&1; cat $O/fetch.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in base line lineNT; do
  ISSCABAC_LIB=$PWD/tools/ab/variant_$v.so ISSCABAC_BENCH_FULL_CAP=1 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/pmc/$v/GRBM -- python3 bench.py --steps 1 --warmup 1 --cpu-seconds 0 --no-verify > $O/pmc/$v.GRBM.log 2>&1
done
python3 - <<'PY' > gpurun_out/r06b/clock.txt 2>&1
import csv, glob, collections
for v in ("base", "line", "lineNT"):
    cnt = collections.defaultdict(list); dur = collections.defaultdict(list)
    for p in glob.glob("gpurun_out/r06b/pmc/%s/GRBM/*/*_counter_collection.csv" % v):
        for r in csv.DictReader(open(p)):
            k = r["Kernel_Name"]
            if "k_encode" in k or "k_decode<" in k or "k_decodeI" in k:
                cnt[k[:40]].append(float(r["Counter_Value"]))
                if "Start_Timestamp" in r: dur[k[:40]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    for k in cnt:
        c = sum(cnt[k]) / len(cnt[k]); d = sum(dur[k]) / len(dur[k]) if dur[k] else 0
        print(v, k, "GRBM_GUI_ACTIVE %.0f  dur %.3f ms  -> %.2f GHz (counter / 8 XCD / time)" % (c, d, c / 8 / (d * 1e-3) / 1e9 if d else 0))
PY
cat gpurun_out/r06b/clock.txt
