#!/bin/bash
# round 5, call e: A's tile path variants (producer at priority 1 in all of them)
set -u
OUT=gpurun_out/r05e; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
LIBS="tools/ab/variant_nopair.so tools/ab/variant_pa1b.so tools/ab/variant_ptp.so tools/ab/variant_pns.so tools/ab/variant_ptpns.so"
for l in tools/ab/variant_ptpns.so; do
  ISSCABAC_LIB=$PWD/$l timeout 300 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu > $OUT/parity_$(basename $l .so).txt 2>&1; echo "parity $l rc=$?" | tee -a $OUT/ab.txt
done
for S in 131072 65536; do
  echo "== $S streams" | tee -a $OUT/ab.txt
  timeout 900 python3 tools/ab_bench.py $LIBS --rounds 3 -- --streams $S --steps 40 --warmup 5 >> $OUT/ab.txt 2>&1
done
cat $OUT/ab.txt
