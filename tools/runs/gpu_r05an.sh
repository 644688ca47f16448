#!/bin/bash
# round 5, call an (absolute table addresses on the chain): batch16 with a straight-line path for pieces of 16 context bins -- parity, soak, rates
set -u
OUT=gpurun_out/r05an; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python3 -m pytest tests/test_gpu_wide.py tests/test_mex_gateway.py -m gpu -x -q > $OUT/gpu_tests.txt 2>&1; tail -3 $OUT/gpu_tests.txt
for C in 999 500 256; do
  timeout 300 python3 tools/wide_rate.py 65536 1024 $C >> $OUT/wide_rate.txt 2>&1
  timeout 300 python3 tools/wide_rate.py 1048576 1024 $C >> $OUT/wide_rate.txt 2>&1
done
timeout 300 python3 tools/wide_rate.py 32768 4096 999 >> $OUT/wide_rate.txt 2>&1
grep streams $OUT/wide_rate.txt
timeout 400 python3 tests/soak/gpu_fuzz_wide.py 120 9 > $OUT/soak_wide.txt 2>&1; tail -2 $OUT/soak_wide.txt
