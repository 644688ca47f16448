#!/bin/bash
# round 6, run x: batch16 kernels start four waves and keep the one on the SIMD a per-CU ticket names -- parity, rates A/B, soak
O=gpurun_out/r06x; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu -k "wide or batch16 or 999" > $O/gputests_wide.txt 2>&1; tail -2 $O/gputests_wide.txt
for r in 1 2; do for lib in tools/ab/variant_base.so isscabac_amd/lib/libisscabac_hip.so; do
  echo "== $lib"; for c in 999 500 256; do ISSCABAC_LIB=$PWD/$lib python tools/wide_rate.py 1048576 1024 $c 2>&1 | grep -v amdgpu; done
  ISSCABAC_LIB=$PWD/$lib python tools/wide_rate.py 65536 1024 999 2>&1 | grep -v amdgpu
done; done > $O/ab_wide_rate.txt 2>&1; cat $O/ab_wide_rate.txt
timeout 200 python tests/soak/gpu_fuzz_wide.py 90 651 > $O/soak_wide.txt 2>&1; tail -1 $O/soak_wide.txt
