#!/bin/bash
# round 5, call b: first run of the producer / consumer encode pair: parity tests (both tunings), sweep A/B against the build without it
set -u
OUT=gpurun_out/r05b; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 120 python3 __graft_entry__.py smoke > $OUT/smoke.txt 2>&1; echo "smoke rc=$?" | tee -a $OUT/rc.txt
timeout 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu > $OUT/gpu_parity.txt 2>&1; echo "parity rc=$?" | tee -a $OUT/rc.txt
tail -5 $OUT/gpu_parity.txt
for l in isscabac_amd/lib/libisscabac_hip.so tools/ab/variant_nopair.so; do
  ISSCABAC_LIB=$PWD/$l timeout 300 python3 tools/s_sweep.py --streams 1048576,262144,131072,65536,32768 --out $OUT/sweep_$(basename $l .so).json > $OUT/sweep_$(basename $l .so).txt 2>&1; echo "sweep $l rc=$?" | tee -a $OUT/rc.txt
  grep streams $OUT/sweep_$(basename $l .so).txt
done
