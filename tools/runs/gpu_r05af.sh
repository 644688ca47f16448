#!/bin/bash
# round 5, call af: the whole GPU suite and the handle / wide / symbol soaks on the build with the live kernel and the LDS batch16
set -u
OUT=gpurun_out/r05af; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 1800 python3 -m pytest tests -m gpu -x -q > $OUT/gpu_tests.txt 2>&1; tail -3 $OUT/gpu_tests.txt
timeout 400 python3 tests/soak/gpu_fuzz_handle.py 150 11 > $OUT/soak_handle.txt 2>&1; tail -1 $OUT/soak_handle.txt
ISSCABAC_STREAM_LIVE=0 timeout 400 python3 tests/soak/gpu_fuzz_handle.py 60 12 > $OUT/soak_handle_nolive.txt 2>&1; tail -1 $OUT/soak_handle_nolive.txt
timeout 400 python3 tests/soak/gpu_fuzz_wide.py 120 5 > $OUT/soak_wide.txt 2>&1; tail -1 $OUT/soak_wide.txt
timeout 400 python3 tests/soak/gpu_fuzz_symbols.py 120 6 > $OUT/soak_symbols.txt 2>&1; tail -1 $OUT/soak_symbols.txt
timeout 300 python3 bench.py --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err; cat $OUT/bench.json | cut -c1-400
