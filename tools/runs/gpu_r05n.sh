#!/bin/bash
# round 5, call n: bank-partitioned replicas of the decoder's entry table: time and LDS conflicts
set -u
OUT=gpurun_out/r05n; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ISSCABAC_LIB=$PWD/tools/ab/variant_tabrepl.so timeout 300 python3 bench.py --steps 3 --warmup 1 --cpu-seconds 3 > $OUT/verify.json 2> $OUT/verify.err; echo "verify rc=$?" | tee $OUT/ab.txt
timeout 900 python3 tools/ab_bench.py isscabac_amd/lib/libisscabac_hip.so tools/ab/variant_tabrepl.so --rounds 6 >> $OUT/ab.txt 2>&1
for l in isscabac_amd/lib/libisscabac_hip.so tools/ab/variant_tabrepl.so; do
  n=$(basename $l .so)
  ISSCABAC_LIB=$PWD/$l timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES --output-format csv -d $OUT/pmc_$n/sq -- python3 bench.py --steps 2 --warmup 1 --cpu-seconds 0 --no-verify > $OUT/pmc_$n.log 2>&1
  python3 tools/pmc_summary.py $OUT/pmc_$n | python3 -c "
import json,sys
d=json.load(sys.stdin)
for k,x in d.items():
    if k.startswith('k_decode<'):
        print('$n', k, 'LDS conflict cycles / LDS active cycles = %.1f %%' % (100*x['SQ_LDS_BANK_CONFLICT']/x['SQ_LDS_IDX_ACTIVE']), 'LDS instructions %.3g' % x['SQ_INSTS_LDS'])
" >> $OUT/ab.txt
done
cat $OUT/ab.txt
