#!/bin/bash
# round 4, third GPU call: A/B + line fills (FETCH_SIZE) of the memory-side experiments
set -u
OUT=gpurun_out/r04c; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
LIBS="tools/ab/variant_shl16_pick2.so tools/ab/variant_p2_lprbc8.so tools/ab/variant_p2_lprc8.so tools/ab/variant_p2_binaddr.so tools/ab/variant_p2_pf48.so tools/ab/variant_p2_pf40.so"
timeout 1500 python3 tools/ab_bench.py $LIBS --rounds 5 > $OUT/ab.txt 2>&1
cat $OUT/ab.txt
bash tools/pmc_fetch.sh $OUT/pmc $LIBS > $OUT/fetch.txt 2>&1
cat $OUT/fetch.txt
for l in tools/ab/variant_p2_binaddr.so tools/ab/variant_p2_pf48.so tools/ab/variant_p2_lprbc8.so; do
  ISSCABAC_LIB=$PWD/$l timeout 300 python3 bench.py --steps 2 --warmup 1 --cpu-seconds 4 > $OUT/verify_$(basename $l .so).json 2> $OUT/verify_$(basename $l .so).err
  echo "$l rc=$?" >> $OUT/verify.txt
done
cat $OUT/verify.txt
