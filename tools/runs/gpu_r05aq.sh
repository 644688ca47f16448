#!/bin/bash
# round 5, call aq: a whole matrix as ONE stream -- columns walked in parallel, the bin sequence through the bin-level encoder
set -u
OUT=gpurun_out/r05aq; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 300 python3 tools/iss_matrix_rate.py > $OUT/iss_matrix_rate.txt 2>&1; grep -v amdgpu $OUT/iss_matrix_rate.txt
ISSCABAC_SYM_NO_MATRIX_SPLIT=1 timeout 300 python3 tools/iss_matrix_rate.py > $OUT/iss_matrix_rate_fused.txt 2>&1; grep -v amdgpu $OUT/iss_matrix_rate_fused.txt
timeout 1500 python3 -m pytest tests -m gpu -x -q > $OUT/gpu_tests.txt 2>&1; tail -3 $OUT/gpu_tests.txt
timeout 400 python3 tests/soak/gpu_fuzz_symbols.py 120 31 > $OUT/soak_symbols.txt 2>&1; tail -1 $OUT/soak_symbols.txt
