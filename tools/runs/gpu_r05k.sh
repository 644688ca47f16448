#!/bin/bash
# round 5, call k: the decoder pair at 64 Ki / 128 Ki streams with the helper's tile by lane pairs and the decoding wave preferred throughout;
# the gateway tests with uint16 ctx
set -u
OUT=gpurun_out/r05k; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 600 python3 -m pytest tests/test_mex_gateway.py tests/test_gpu_wide.py -x -q -m gpu > $OUT/gw_tests.txt 2>&1; echo "gateway rc=$?" | tee -a $OUT/rc.txt; tail -3 $OUT/gw_tests.txt
LIBS="isscabac_amd/lib/libisscabac_hip.so tools/ab/variant_dp8.so tools/ab/variant_dp8tp.so tools/ab/variant_dp8pr.so tools/ab/variant_dp8all.so"
for S in 131072 65536 32768; do
  echo "== $S streams" | tee -a $OUT/ab.txt
  timeout 900 python3 tools/ab_bench.py $LIBS --rounds 3 -- --streams $S --steps 40 --warmup 5 >> $OUT/ab.txt 2>&1
done
cat $OUT/ab.txt
