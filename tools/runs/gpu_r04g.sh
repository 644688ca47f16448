#!/bin/bash
# round 4: paired-lane tile mapping (two full ds_write_b128 per part instead of four / eight partial ones): A/B, fills, verify
set -u
OUT=gpurun_out/r04g; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
LIBS="tools/ab/variant_final.so tools/ab/variant_pair_enc.so tools/ab/variant_pair_dec.so tools/ab/variant_pair_both.so"
for l in tools/ab/variant_pair_both.so; do
  ISSCABAC_LIB=$PWD/$l timeout 300 python3 bench.py --steps 2 --warmup 1 --cpu-seconds 4 > $OUT/verify_$(basename $l .so).json 2> $OUT/verify_$(basename $l .so).err
  echo "$l rc=$?" >> $OUT/verify.txt
done
cat $OUT/verify.txt
timeout 1500 python3 tools/ab_bench.py $LIBS --rounds 6 > $OUT/ab.txt 2>&1
cat $OUT/ab.txt
bash tools/pmc_fetch.sh $OUT/pmc tools/ab/variant_final.so tools/ab/variant_pair_both.so > $OUT/fetch.txt 2>&1
cat $OUT/fetch.txt
