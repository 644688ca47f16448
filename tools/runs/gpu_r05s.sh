#!/bin/bash
# round 5: randomised soaks of the final build against the oracle (GPU box)
set -u
OUT=gpurun_out/r05s; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 400 python3 tests/soak/gpu_fuzz.py 240 501 > $OUT/soak_bins.txt 2>&1; tail -2 $OUT/soak_bins.txt
timeout 300 python3 tests/soak/gpu_fuzz_bucketed.py 180 502 > $OUT/soak_bucketed.txt 2>&1; tail -2 $OUT/soak_bucketed.txt
timeout 300 python3 tests/soak/gpu_fuzz_symbols.py 180 503 > $OUT/soak_symbols.txt 2>&1; tail -2 $OUT/soak_symbols.txt
timeout 300 python3 tests/soak/gpu_fuzz_multi.py 180 504 > $OUT/soak_multi.txt 2>&1; tail -2 $OUT/soak_multi.txt
timeout 200 python3 tests/soak/gpu_fuzz_handle.py 90 505 > $OUT/soak_handle.txt 2>&1; tail -2 $OUT/soak_handle.txt
