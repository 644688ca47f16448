#!/bin/bash
set -u
OUT=gpurun_out/r04y; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ISSCABAC_LIB=$PWD/tools/ab/variant_splitf.so timeout 300 python3 bench.py --steps 2 --warmup 1 --cpu-seconds 4 > $OUT/verify_splitf.json 2> $OUT/verify_splitf.err; echo "splitf rc=$?"
timeout 1800 python3 tools/ab_bench.py tools/ab/variant_final.so tools/ab/variant_splitf.so tools/ab/variant_x_noload.so tools/ab/variant_x_nostore.so --rounds 6 > $OUT/ab.txt 2>&1
cat $OUT/ab.txt
