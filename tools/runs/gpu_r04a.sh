#!/bin/bash
# round 4, first GPU call: microbenchmarks (half-EXEC issue, more VALU op costs), FETCH_SIZE / WRITE_SIZE calibration on known
# byte counts, A/B of the experiment builds, the GPU test suite.  usage (GPU box): bash tools/gpu_r04a.sh
set -u
OUT=gpurun_out/r04a; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 120 tools/ubench/valu_half > $OUT/ubench_valu_half.txt 2>&1
timeout 300 tools/ubench/valu_ops2 > $OUT/ubench_valu_ops2.txt 2>&1
timeout 120 tools/ubench/fetch_calib > $OUT/fetch_calib_plain.txt 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/calib/$c -- tools/ubench/fetch_calib > $OUT/calib_$c.log 2>&1
done
python3 tools/pmc_summary.py $OUT/calib $OUT/calib_summary.json > /dev/null 2>&1
timeout 1200 python3 tools/ab_bench.py tools/ab/base_r03.so isscabac_amd/lib/libisscabac_hip.so tools/ab/variant_pick2.so tools/ab/variant_mul24.so tools/ab/variant_gather32.so tools/ab/variant_gather8.so tools/ab/variant_combo.so --rounds 3 > $OUT/ab.txt 2>&1
# the variants must still be bit-exact: one verified step each
for l in tools/ab/variant_pick2.so tools/ab/variant_mul24.so tools/ab/variant_combo.so; do
  ISSCABAC_LIB=$PWD/$l timeout 300 python3 bench.py --steps 2 --warmup 1 --cpu-seconds 4 > $OUT/verify_$(basename $l .so).json 2> $OUT/verify_$(basename $l .so).err
  echo "$l rc=$?" >> $OUT/verify.txt
done
timeout 900 python3 -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.txt 2>&1
tail -3 $OUT/pytest_gpu.txt
cat $OUT/ab.txt
