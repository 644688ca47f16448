#!/bin/bash
# round 4, second GPU call: what a miss fetches (partial-line patterns of fetch_calib), A/B of the 16-bit shifts and the mask pick,
# the GPU suite on the candidate build.  usage (GPU box): bash tools/gpu_r04b.sh
set -u
OUT=gpurun_out/r04b; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 120 tools/ubench/fetch_calib > $OUT/fetch_calib_plain.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/calib/FETCH_SIZE -- tools/ubench/fetch_calib > $OUT/calib_FETCH_SIZE.log 2>&1
python3 tools/pmc_summary.py $OUT/calib $OUT/calib_summary.json > /dev/null 2>&1
timeout 1500 python3 tools/ab_bench.py tools/ab/base_r03.so tools/ab/variant_noshl16.so tools/ab/variant_shl16.so tools/ab/variant_shl16_pick2.so --rounds 6 > $OUT/ab.txt 2>&1
ISSCABAC_LIB=$PWD/tools/ab/variant_shl16_pick2.so timeout 900 python3 -m pytest tests -x -q -m gpu > $OUT/pytest_gpu_pick2.txt 2>&1
tail -3 $OUT/pytest_gpu_pick2.txt
timeout 900 python3 -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.txt 2>&1
tail -3 $OUT/pytest_gpu.txt
cat $OUT/ab.txt
