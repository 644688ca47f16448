#!/bin/bash
# round 4, fifth GPU call: GPU suite on the build with the host symbol calls on the pinned ring + the fused small scan/gather;
# host symbol API rates before (round 3's library) and after
set -u
OUT=gpurun_out/r04e; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 1200 python3 -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.txt 2>&1
tail -5 $OUT/pytest_gpu.txt
timeout 900 python3 tests/perf/symbol_host_rate.py > $OUT/symbol_host_rate.txt 2>&1
cat $OUT/symbol_host_rate.txt
ISSCABAC_LIB=$PWD/tools/ab/base_r03.so timeout 900 python3 tests/perf/symbol_host_rate.py --no-gateway > $OUT/symbol_host_rate_r03_library.txt 2>&1
cat $OUT/symbol_host_rate_r03_library.txt
timeout 300 python3 tools/symbol_rate.py > $OUT/symbol_rate.txt 2>&1
cat $OUT/symbol_rate.txt
timeout 300 python3 tools/per_bin_rate.py > $OUT/per_bin_rate.txt 2>&1
tail -5 $OUT/per_bin_rate.txt
