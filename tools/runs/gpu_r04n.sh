#!/bin/bash
set -u
OUT=gpurun_out/r04n; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 1500 python3 tools/ab_bench.py tools/ab/variant_mixed1.so tools/ab/variant_mixed2.so tools/ab/variant_final.so --rounds 8 > $OUT/ab_config3.txt 2>&1
cat $OUT/ab_config3.txt
