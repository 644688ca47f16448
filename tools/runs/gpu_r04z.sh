#!/bin/bash
# round 4: does a fifth encoder wave per SIMD help?  (2 KiB state image: 32-context data; timing-only ablations beside the real build)
set -u
OUT=gpurun_out/r04z; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in final sb32 x_nomem x_nomem_sb32 x_nomem_step3 x_nomem_step3_sb32 x_nomem_step1 x_nomem_step1_sb32; do
  for rep in 1 2; do
    echo -n "$v: " >> $OUT/ctx32.txt
    ISSCABAC_NO_VERIFY=1 ISSCABAC_LIB=$PWD/tools/ab/variant_$v.so timeout 200 python3 tools/ctx32_rate.py 2>&1 | grep -v amdgpu.ids >> $OUT/ctx32.txt
  done
done
cat $OUT/ctx32.txt
