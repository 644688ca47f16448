#!/bin/bash
# round 4: which half of the encoder's step bounds it?  timing-only ablation builds (wrong output; only the encode time counts)
set -u
OUT=gpurun_out/r04x; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 1800 python3 tools/ab_bench.py tools/ab/variant_final.so tools/ab/variant_x_nomem.so tools/ab/variant_x_nomem_step1.so tools/ab/variant_x_nomem_step2.so tools/ab/variant_x_nomem_step3.so tools/ab/variant_x_nomem_noflush.so tools/ab/variant_x_step3.so tools/ab/variant_x_noflush.so --rounds 4 > $OUT/ab.txt 2>&1
cat $OUT/ab.txt
