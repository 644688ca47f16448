#!/bin/bash
# round 6, run v: waves where the SIMDs are (k_encode_pair pinned through four-wave launches; four-wave workgroups above 64 contexts) -- parity, sweeps A/B
O=gpurun_out/r06v; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/gputests.txt 2>&1; tail -3 $O/gputests.txt
for r in 1 2; do
  echo "== base"; ISSCABAC_LIB=$PWD/tools/ab/variant_base.so python tools/s_sweep.py --streams 1048576,262144,131072,65536,32768 2>&1 | grep -v amdgpu
  echo "== waves placed"; python tools/s_sweep.py --streams 1048576,262144,131072,65536,32768 2>&1 | grep -v amdgpu
done > $O/ab_s_sweep.txt 2>&1; cat $O/ab_s_sweep.txt
for r in 1 2; do
  echo "== base"; ISSCABAC_LIB=$PWD/tools/ab/variant_base.so python tools/ctx_rates.py 2>&1 | grep -v amdgpu
  echo "== waves placed"; python tools/ctx_rates.py 2>&1 | grep -v amdgpu
done > $O/ab_ctx_rates.txt 2>&1; cat $O/ab_ctx_rates.txt
timeout 200 python tests/soak/gpu_fuzz.py 100 641 > $O/soak.txt 2>&1; tail -1 $O/soak.txt
timeout 200 python tests/soak/gpu_fuzz_bucketed.py 60 642 > $O/soak_bucketed.txt 2>&1; tail -1 $O/soak_bucketed.txt
