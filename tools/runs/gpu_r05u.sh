#!/bin/bash
# round 5, call u: the latency decoders' decision chain with one multiply-add (ISSCABAC_DEC_LAT_MAD): parity, then A/B
set -u
OUT=gpurun_out/r05u; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ISSCABAC_LIB=$PWD/tools/ab/variant_latmad.so timeout 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu --deselect tests/test_gpu_parity.py::test_native_library_is_loaded > $OUT/parity.txt 2>&1; echo "parity rc=$?" | tee $OUT/ab.txt; tail -2 $OUT/parity.txt
LIBS="isscabac_amd/lib/libisscabac_hip.so tools/ab/variant_latmad.so"
for S in 131072 65536 16384; do
  echo "== $S streams" | tee -a $OUT/ab.txt
  timeout 900 python3 tools/ab_bench.py $LIBS --rounds 5 -- --streams $S --steps 40 --warmup 5 >> $OUT/ab.txt 2>&1
done
cat $OUT/ab.txt
