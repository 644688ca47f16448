#!/bin/bash
# round 5, call f: the pair as built (producer priority 1, paired-lane tiles): whole GPU suite, sweep, config-5-like data at small S
set -u
OUT=gpurun_out/r05f; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 600 python3 -m pytest tests -x -q -m gpu > $OUT/gpu_tests.txt 2>&1; echo "pytest rc=$?" | tee -a $OUT/rc.txt; tail -3 $OUT/gpu_tests.txt
timeout 300 python3 tools/s_sweep.py --out $OUT/s_sweep.json > $OUT/s_sweep.txt 2>&1; grep streams $OUT/s_sweep.txt
LIBS="isscabac_amd/lib/libisscabac_hip.so tools/ab/variant_nopair.so"
for S in 131072 65536; do
  echo "== config5 data, $S streams" | tee -a $OUT/ab.txt
  timeout 600 python3 tools/ab_bench.py $LIBS --rounds 3 --workload config5_per_gpu -- --streams $S --steps 20 --warmup 3 >> $OUT/ab.txt 2>&1
done
echo "== config2" | tee -a $OUT/ab.txt
timeout 600 python3 tools/ab_bench.py $LIBS --rounds 3 --workload config2 -- --steps 100 --warmup 5 >> $OUT/ab.txt 2>&1
cat $OUT/ab.txt
