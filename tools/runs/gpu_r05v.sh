#!/bin/bash
# round 5, call v: longer soaks of the final build (the GPU minutes are there), the batch16 soak among them
set -u
OUT=gpurun_out/r05v; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 400 python3 tests/soak/gpu_fuzz_wide.py 240 511 > $OUT/soak_wide.txt 2>&1; tail -2 $OUT/soak_wide.txt
timeout 800 python3 tests/soak/gpu_fuzz.py 600 512 > $OUT/soak_bins.txt 2>&1; tail -2 $OUT/soak_bins.txt
timeout 500 python3 tests/soak/gpu_fuzz_bucketed.py 300 513 > $OUT/soak_bucketed.txt 2>&1; tail -2 $OUT/soak_bucketed.txt
timeout 500 python3 tests/soak/gpu_fuzz_symbols.py 300 514 > $OUT/soak_symbols.txt 2>&1; tail -2 $OUT/soak_symbols.txt
timeout 400 python3 tests/soak/gpu_fuzz_multi.py 240 515 > $OUT/soak_multi.txt 2>&1; tail -2 $OUT/soak_multi.txt
