#!/bin/bash
# round 5, call am: lanes per stream of k_gather (16 in the tree) -- 8 / 32 / 64
set -u
OUT=gpurun_out/r05am; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
LIBS="isscabac_amd/lib/libisscabac_hip.so tools/ab/variant_l8.so tools/ab/variant_l32.so tools/ab/variant_l64.so"
timeout 900 python3 tools/ab_bench.py $LIBS --rounds 5 > $OUT/ab.txt 2>&1
timeout 900 python3 tools/ab_bench.py $LIBS --rounds 3 --workload config5_per_gpu >> $OUT/ab.txt 2>&1
grep -v amdgpu $OUT/ab.txt
