#!/bin/bash
# round 5, call t: the final build: whole GPU suite, the sweep once more
set -u
OUT=gpurun_out/r05t; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python3 -m pytest tests -x -q -m gpu > $OUT/gpu_tests.txt 2>&1; echo "pytest rc=$?" | tee -a $OUT/rc.txt; tail -4 $OUT/gpu_tests.txt
timeout 120 python3 __graft_entry__.py smoke > $OUT/smoke.txt 2>&1; echo "smoke rc=$?" | tee -a $OUT/rc.txt; tail -1 $OUT/smoke.txt
timeout 300 python3 tools/s_sweep.py --streams 1048576,131072,65536,16384 > $OUT/s_sweep.txt 2>&1; grep streams $OUT/s_sweep.txt
