#!/bin/bash
# round 5, call ar: the empirical context initialisation counted with wave ballots and a grid-stride loop
set -u
OUT=gpurun_out/r05ar; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python3 -m pytest tests -m gpu -x -q -k "iss or init or symbol or matrix or coder or gateway" > $OUT/gpu_tests.txt 2>&1; tail -2 $OUT/gpu_tests.txt
timeout 900 python3 tests/perf/symbol_host_rate.py --no-gateway > $OUT/symbol_host_rate.txt 2>&1; grep "columns x" $OUT/symbol_host_rate.txt | cut -c1-260
timeout 400 python3 tests/soak/gpu_fuzz_symbols.py 90 41 > $OUT/soak_symbols.txt 2>&1; tail -1 $OUT/soak_symbols.txt
