#!/bin/bash
# round 5, call i: symbol host calls of small batches through the pinned block; decode-pair threshold
set -u
OUT=gpurun_out/r05i; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python3 -m pytest tests -x -q -m gpu > $OUT/gpu_tests.txt 2>&1; echo "pytest rc=$?" | tee -a $OUT/rc.txt; tail -5 $OUT/gpu_tests.txt
timeout 600 python3 tests/perf/symbol_host_rate.py --real-only > $OUT/symbol_host_rate.txt 2>&1; echo "symrate rc=$?" | tee -a $OUT/rc.txt
cat $OUT/symbol_host_rate.txt | cut -c1-400
timeout 300 python3 tools/s_sweep.py --streams 1048576,131072,65536,49152,32768,16384 --out $OUT/s_sweep.json > $OUT/s_sweep.txt 2>&1; grep streams $OUT/s_sweep.txt
