#!/bin/bash
# round 5, call j: batches of more than 255 contexts (16-bit indices)
set -u
OUT=gpurun_out/r05j; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 600 python3 -m pytest tests/test_gpu_wide.py -x -q > $OUT/wide_tests.txt 2>&1; echo "wide tests rc=$?" | tee -a $OUT/rc.txt; tail -15 $OUT/wide_tests.txt
timeout 300 python3 tools/wide_rate.py 65536 1024 999 > $OUT/wide_rate.txt 2>&1
timeout 300 python3 tools/wide_rate.py 1048576 1024 999 >> $OUT/wide_rate.txt 2>&1
timeout 300 python3 tools/wide_rate.py 1048576 1024 300 >> $OUT/wide_rate.txt 2>&1
cat $OUT/wide_rate.txt
