#!/bin/bash
# round 5, call ah (EXP 5 two deep in flight): how much is there to gain on the producer's chain?  timing-only builds (wrong bytes): EXP 5 = no forwarding
# (every entry lookup on the state read ahead: no chain at all), EXP 1 = B skips its arithmetic
set -u
OUT=gpurun_out/r05ah; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
LIBS="isscabac_amd/lib/libisscabac_hip.so tools/ab/variant_exp5.so tools/ab/variant_exp1.so"
for S in 131072 65536 16384; do
  echo "== $S streams" | tee -a $OUT/ab.txt
  timeout 900 python3 tools/ab_bench.py $LIBS --rounds 3 -- --streams $S --steps 40 --warmup 5 >> $OUT/ab.txt 2>&1
done
cat $OUT/ab.txt
