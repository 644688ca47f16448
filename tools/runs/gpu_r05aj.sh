#!/bin/bash
# round 5, call aj: how much could a speculative state fetch buy the latency decoder?  timing-only build ISSCABAC_DEC_LAT_EXP=1:
# the next bin's entry lookup issued before this bin's decision (no forwarding: wrong for repeated contexts)
set -u
OUT=gpurun_out/r05aj; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
LIBS="isscabac_amd/lib/libisscabac_hip.so tools/ab/variant_dexp.so"
for S in 131072 65536 49152; do
  echo "== $S streams" | tee -a $OUT/ab.txt
  timeout 900 python3 tools/ab_bench.py $LIBS --rounds 3 -- --streams $S --steps 40 --warmup 5 >> $OUT/ab.txt 2>&1
done
grep -v amdgpu $OUT/ab.txt
