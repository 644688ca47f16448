#!/bin/bash
# usage: bash tools/pmc_fetch.sh <outdir> <lib.so>...   -- FETCH_SIZE / WRITE_SIZE per kernel for each build (GPU box)
set -u
OUT=$1; shift; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for lib in "$@"; do
  name=$(basename $lib .so)
  for c in FETCH_SIZE WRITE_SIZE; do
    ISSCABAC_LIB=$PWD/$lib ISSCABAC_BENCH_FULL_CAP=1 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/$name/$c -- python3 bench.py --steps 1 --warmup 1 --cpu-seconds 0 --no-verify > $OUT/$name.$c.log 2>&1
  done
  python3 tools/pmc_summary.py $OUT/$name | python3 -c "
import json,sys
d=json.load(sys.stdin)
for k,x in d.items():
    if 'encode' in k or 'decode' in k:
        print('$name %-14s FETCH %.2f GB WRITE %.2f GB' % (k, x.get('FETCH_SIZE',0)*1024/1e9, x.get('WRITE_SIZE',0)*1024/1e9))
"
done
