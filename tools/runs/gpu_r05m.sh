#!/bin/bash
# round 5, call m: producer role alternating between successive workgroups
set -u
OUT=gpurun_out/r05m; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
LIBS="isscabac_amd/lib/libisscabac_hip.so tools/ab/variant_altroles.so"
for S in 131072 65536; do
  echo "== $S streams" | tee -a $OUT/ab.txt
  timeout 900 python3 tools/ab_bench.py $LIBS --rounds 7 -- --streams $S --steps 40 --warmup 5 >> $OUT/ab.txt 2>&1
done
cat $OUT/ab.txt
