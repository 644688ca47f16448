#!/bin/bash
# round 5, call x: asynchronous symbol entry points -- the GPU suite, the symbol kernels' rate with steps chained, config 4 at its real size
set -u
OUT=gpurun_out/r05x; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 1500 python3 -m pytest tests -m gpu -x -q > $OUT/gpu_tests.txt 2>&1; tail -3 $OUT/gpu_tests.txt
timeout 600 python3 tools/symbol_rate.py > $OUT/symbol_rate.txt 2>&1; cat $OUT/symbol_rate.txt
timeout 600 python3 tests/perf/symbol_host_rate.py --real-only > $OUT/symbol_host_real.txt 2>&1; cat $OUT/symbol_host_real.txt
