#!/bin/bash
# round 5, call c: the encode pair taken apart (timing-only builds)
set -u
OUT=gpurun_out/r05c; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
LIBS="isscabac_amd/lib/libisscabac_hip.so tools/ab/variant_nopair.so tools/ab/variant_px1.so tools/ab/variant_px2.so tools/ab/variant_px3.so tools/ab/variant_pprio.so"
for S in 65536 131072 16384; do
  echo "== $S streams" | tee -a $OUT/ab.txt
  timeout 900 python3 tools/ab_bench.py $LIBS --rounds 3 -- --streams $S >> $OUT/ab.txt 2>&1
done
cat $OUT/ab.txt
