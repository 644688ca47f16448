#!/bin/bash
# round 5, call al: k_gather moving two neighbouring streams per lane group (loads of both in flight) against one
set -u
OUT=gpurun_out/r05al; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "not handle and not symbol and not mex" > $OUT/tests.txt 2>&1; tail -2 $OUT/tests.txt
LIBS="isscabac_amd/lib/libisscabac_hip.so tools/ab/variant_g1.so"
timeout 900 python3 tools/ab_bench.py $LIBS --rounds 5 > $OUT/ab.txt 2>&1
timeout 900 python3 tools/ab_bench.py $LIBS --rounds 3 --workload config5_per_gpu >> $OUT/ab.txt 2>&1
grep -v amdgpu $OUT/ab.txt
