#!/bin/bash
# round 5, call ak: long soaks of the round's last build (live kernel, LDS batch16 with the straight-line path, asynchronous symbol
# calls, statistics counters in the LDS) -- every soak, new seeds
set -u
OUT=gpurun_out/r05ak; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 500 python3 tests/soak/gpu_fuzz_wide.py 300 611 > $OUT/soak_wide.txt 2>&1; tail -1 $OUT/soak_wide.txt
timeout 500 python3 tests/soak/gpu_fuzz_handle.py 300 612 > $OUT/soak_handle.txt 2>&1; tail -1 $OUT/soak_handle.txt
timeout 500 python3 tests/soak/gpu_fuzz_symbols.py 300 613 > $OUT/soak_symbols.txt 2>&1; tail -1 $OUT/soak_symbols.txt
timeout 500 python3 tests/soak/gpu_fuzz.py 300 614 > $OUT/soak_bins.txt 2>&1; tail -1 $OUT/soak_bins.txt
timeout 400 python3 tests/soak/gpu_fuzz_bucketed.py 180 615 > $OUT/soak_bucketed.txt 2>&1; tail -1 $OUT/soak_bucketed.txt
timeout 400 python3 tests/soak/gpu_fuzz_multi.py 180 616 > $OUT/soak_multi.txt 2>&1; tail -1 $OUT/soak_multi.txt
timeout 300 python3 tests/soak/long_streams.py > $OUT/long_streams.txt 2>&1; tail -2 $OUT/long_streams.txt
