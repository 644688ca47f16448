#!/bin/bash
# round 6, call s: long soaks of the round's last build -- every soak, new seeds
set -u
OUT=gpurun_out/r06s; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 500 python3 tests/soak/gpu_fuzz.py 360 621 > $OUT/soak_bins.txt 2>&1; tail -1 $OUT/soak_bins.txt
timeout 400 python3 tests/soak/gpu_fuzz_bucketed.py 240 622 > $OUT/soak_bucketed.txt 2>&1; tail -1 $OUT/soak_bucketed.txt
timeout 300 python3 tests/soak/gpu_fuzz_symbols.py 150 623 > $OUT/soak_symbols.txt 2>&1; tail -1 $OUT/soak_symbols.txt
timeout 300 python3 tests/soak/gpu_fuzz_wide.py 120 624 > $OUT/soak_wide.txt 2>&1; tail -1 $OUT/soak_wide.txt
timeout 300 python3 tests/soak/gpu_fuzz_handle.py 120 625 > $OUT/soak_handle.txt 2>&1; tail -1 $OUT/soak_handle.txt
timeout 300 python3 tests/soak/gpu_fuzz_multi.py 120 626 > $OUT/soak_multi.txt 2>&1; tail -1 $OUT/soak_multi.txt
timeout 300 python3 tests/soak/long_streams.py > $OUT/long_streams.txt 2>&1; tail -2 $OUT/long_streams.txt
