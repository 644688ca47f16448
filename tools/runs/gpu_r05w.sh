#!/bin/bash
# round 5, call w: the decoder pair's launch-size threshold (2 / 3 / 4 waves of streams per CU) and helper role alternating between workgroups
set -u
OUT=gpurun_out/r05w; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
LIBS="isscabac_amd/lib/libisscabac_hip.so tools/ab/variant_dp3.so tools/ab/variant_dp4.so tools/ab/variant_dp4alt.so"
for S in 65536 49152 40960; do
  echo "== $S streams" | tee -a $OUT/ab.txt
  timeout 900 python3 tools/ab_bench.py $LIBS --rounds 5 -- --streams $S --steps 40 --warmup 5 >> $OUT/ab.txt 2>&1
done
cat $OUT/ab.txt
