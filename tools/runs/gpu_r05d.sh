#!/bin/bash
# round 5, call d: producer priority, consumer alone
set -u
OUT=gpurun_out/r05d; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
LIBS="isscabac_amd/lib/libisscabac_hip.so tools/ab/variant_nopair.so tools/ab/variant_pa1.so tools/ab/variant_pa3.so tools/ab/variant_px1.so tools/ab/variant_px1a3.so tools/ab/variant_px4.so"
for S in 131072 65536 16384; do
  echo "== $S streams" | tee -a $OUT/ab.txt
  timeout 900 python3 tools/ab_bench.py $LIBS --rounds 3 -- --streams $S --steps 40 --warmup 5 >> $OUT/ab.txt 2>&1
done
cat $OUT/ab.txt
