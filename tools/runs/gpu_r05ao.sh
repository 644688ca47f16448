#!/bin/bash
# round 5, call ao: live kernels leave before anything else is enqueued for their engine (registry + ENGINE_ENTER), idle time 200 us
set -u
OUT=gpurun_out/r05ao; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export LD_LIBRARY_PATH=/opt/rocm/lib:${LD_LIBRARY_PATH:-}
timeout 600 python3 tools/live_interleave.py > $OUT/live_interleave.txt 2>&1; grep -v amdgpu $OUT/live_interleave.txt
timeout 300 tests/_build/per_bin_native 200000 > $OUT/per_bin_native.txt 2>&1; cat $OUT/per_bin_native.txt
timeout 900 python3 -m pytest tests -m gpu -x -q -k "handle or stream or mex or gateway or f4 or trace or example" > $OUT/gpu_tests_handle.txt 2>&1; tail -2 $OUT/gpu_tests_handle.txt
timeout 300 python3 tools/per_bin_rate.py > $OUT/per_bin_rate.txt 2>&1; grep -v amdgpu $OUT/per_bin_rate.txt
timeout 300 python3 tests/soak/gpu_fuzz_handle.py 90 21 > $OUT/soak_handle.txt 2>&1; tail -1 $OUT/soak_handle.txt
