#!/bin/bash
# round 5, call y: statistics counters in the LDS, the small pinned block with the statistics -- the GPU suite, config 4 through the host calls
set -u
OUT=gpurun_out/r05y; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 1500 python3 -m pytest tests -m gpu -x -q > $OUT/gpu_tests.txt 2>&1; tail -3 $OUT/gpu_tests.txt
timeout 600 python3 tests/perf/symbol_host_rate.py --real-only > $OUT/symbol_host_real.txt 2>&1; cat $OUT/symbol_host_real.txt
timeout 900 python3 tests/perf/symbol_host_rate.py --no-gateway > $OUT/symbol_host_rate.txt 2>&1; cat $OUT/symbol_host_rate.txt
timeout 600 python3 tests/soak/gpu_fuzz_symbols.py 60 5 > $OUT/soak_symbols.txt 2>&1; tail -3 $OUT/soak_symbols.txt
