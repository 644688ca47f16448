#!/bin/bash
# round 5, call ae (states staged in the LDS): per-bin handle served by a live kernel -- native per-call times with and without it, handle tests, handle soak
set -u
OUT=gpurun_out/r05ae; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export LD_LIBRARY_PATH=/opt/rocm/lib:${LD_LIBRARY_PATH:-}
echo "== live kernel on" > $OUT/per_bin_native.txt
timeout 300 tests/_build/per_bin_native 200000 >> $OUT/per_bin_native.txt 2>&1; echo "rc $?" >> $OUT/per_bin_native.txt
echo "== ISSCABAC_STREAM_LIVE=0 (a launch + a synchronisation per call)" >> $OUT/per_bin_native.txt
ISSCABAC_STREAM_LIVE=0 timeout 300 tests/_build/per_bin_native 200000 >> $OUT/per_bin_native.txt 2>&1; echo "rc $?" >> $OUT/per_bin_native.txt
cat $OUT/per_bin_native.txt
timeout 900 python3 -m pytest tests -m gpu -x -q -k "handle or stream or mex or gateway or f4 or trace" > $OUT/gpu_tests_handle.txt 2>&1; tail -3 $OUT/gpu_tests_handle.txt
timeout 300 python3 tools/per_bin_rate.py > $OUT/per_bin_rate.txt 2>&1; cat $OUT/per_bin_rate.txt
timeout 300 python3 tests/soak/gpu_fuzz_handle.py 60 7 > $OUT/soak_handle.txt 2>&1; tail -2 $OUT/soak_handle.txt
