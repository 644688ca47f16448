#!/bin/bash
set -u
OUT=gpurun_out/r04f; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python3 tests/perf/symbol_host_rate.py > $OUT/symbol_host_rate.txt 2>&1
cat $OUT/symbol_host_rate.txt
ISSCABAC_LIB=$PWD/tools/ab/base_r03.so timeout 900 python3 tests/perf/symbol_host_rate.py --no-gateway > $OUT/symbol_host_rate_r03_library.txt 2>&1
cat $OUT/symbol_host_rate_r03_library.txt
