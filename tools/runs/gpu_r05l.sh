#!/bin/bash
# round 5, call l: two-launch scan + gather for medium batches, size word through pinned memory, no memset per decode
set -u
OUT=gpurun_out/r05l; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python3 -m pytest tests -x -q -m gpu > $OUT/gpu_tests.txt 2>&1; echo "pytest rc=$?" | tee -a $OUT/rc.txt; tail -4 $OUT/gpu_tests.txt
i=0
for l in isscabac_amd/lib/libisscabac_hip.so tools/ab/variant_head.so isscabac_amd/lib/libisscabac_hip.so tools/ab/variant_head.so; do
  i=$((i+1))
  ISSCABAC_LIB=$PWD/$l timeout 300 python3 tools/s_sweep.py --streams 1048576,262144,131072,65536,32768 > $OUT/sweep_${i}_$(basename $l .so).txt 2>&1; echo "sweep $l rc=$?" | tee -a $OUT/rc.txt
  grep streams $OUT/sweep_${i}_$(basename $l .so).txt
done
