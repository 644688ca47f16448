#!/bin/bash
# round 5, call h: first run of the decoder + helper pair
set -u
OUT=gpurun_out/r05h; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 120 python3 __graft_entry__.py smoke > $OUT/smoke.txt 2>&1; echo "smoke rc=$?" | tee -a $OUT/rc.txt
timeout 900 python3 -m pytest tests -x -q -m gpu > $OUT/gpu_tests.txt 2>&1; echo "pytest rc=$?" | tee -a $OUT/rc.txt; tail -5 $OUT/gpu_tests.txt
for l in isscabac_amd/lib/libisscabac_hip.so tools/ab/variant_nodpair.so; do
  ISSCABAC_LIB=$PWD/$l timeout 300 python3 tools/s_sweep.py --streams 1048576,131072,65536,16384 --out $OUT/sweep_$(basename $l .so).json > $OUT/sweep_$(basename $l .so).txt 2>&1; echo "sweep $l rc=$?" | tee -a $OUT/rc.txt
  grep streams $OUT/sweep_$(basename $l .so).txt
done
