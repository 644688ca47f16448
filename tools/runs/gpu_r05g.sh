#!/bin/bash
# round 5, call g: the pair's mixed path
set -u
OUT=gpurun_out/r05g; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 600 python3 -m pytest tests -x -q -m gpu > $OUT/gpu_tests.txt 2>&1; echo "pytest rc=$?" | tee -a $OUT/rc.txt; tail -3 $OUT/gpu_tests.txt
LIBS="isscabac_amd/lib/libisscabac_hip.so tools/ab/variant_nopair.so"
for S in 131072 65536; do
  echo "== config5 data, $S streams" | tee -a $OUT/ab.txt
  timeout 600 python3 tools/ab_bench.py $LIBS --rounds 3 --workload config5_per_gpu -- --streams $S --steps 20 --warmup 3 >> $OUT/ab.txt 2>&1
done
cat $OUT/ab.txt
