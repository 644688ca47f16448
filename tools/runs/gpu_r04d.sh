#!/bin/bash
# round 4, fourth GPU call: the GPU suite on the candidate build (new one-process multi forms, second-derivation test), A/B with
# more rounds, symbol kernels with either pick, the stream-count sweep
set -u
OUT=gpurun_out/r04d; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 1200 python3 -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.txt 2>&1
tail -5 $OUT/pytest_gpu.txt
timeout 1500 python3 tools/ab_bench.py tools/ab/base_r03.so tools/ab/variant_cand.so tools/ab/variant_cand_dec32.so --rounds 8 > $OUT/ab.txt 2>&1
cat $OUT/ab.txt
for l in tools/ab/variant_cand.so tools/ab/variant_cand_sympick1.so tools/ab/base_r03.so; do
  echo "== $l" >> $OUT/symbol_rate.txt
  ISSCABAC_LIB=$PWD/$l timeout 300 python3 tools/symbol_rate.py >> $OUT/symbol_rate.txt 2>&1
  ISSCABAC_LIB=$PWD/$l timeout 300 python3 tools/symbol_rate.py >> $OUT/symbol_rate.txt 2>&1
done
cat $OUT/symbol_rate.txt
timeout 600 python3 tools/s_sweep.py --out $OUT/s_sweep.json > $OUT/s_sweep.txt 2>&1
cat $OUT/s_sweep.txt
