#!/bin/bash
# usage: bash tools/pmc_valu.sh <outdir> <lib.so>...   -- VALU/SALU/LDS instruction counts per kernel for each build (GPU box)
set -u
OUT=$1; shift; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for lib in "$@"; do
  name=$(basename $lib .so)
  ISSCABAC_LIB=$PWD/$lib ISSCABAC_BENCH_FULL_CAP=1 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES --output-format csv -d $OUT/$name/sq1 -- python3 bench.py --steps 1 --warmup 1 --cpu-seconds 0 --no-verify > $OUT/$name.log 2>&1
  python3 tools/pmc_summary.py $OUT/$name | python3 -c "
import json,sys
d=json.load(sys.stdin)
for k,x in d.items():
    if 'encode' in k or 'decode' in k:
        n=16384*4096
        print('$name %-14s VALU %.2f SALU %.2f LDS %.2f per wave-bin; wave cycles/bin %.0f' % (k, x['SQ_INSTS_VALU']/n, x['SQ_INSTS_SALU']/n, x['SQ_INSTS_LDS']/n, 4*x['SQ_WAVE_CYCLES']/n))
"
done
