#!/bin/bash
# round 6, run u: the pair kernels' waves pinned to SIMDs (four waves start, two stay) -- parity with the pin on, small-launch sweep per pin mode
O=gpurun_out/r06u; mkdir -p $O
ISSCABAC_PAIR_PIN=1 timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > $O/gputests_pin1.txt 2>&1; tail -2 $O/gputests_pin1.txt
ISSCABAC_PAIR_PIN=3 timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "latency" > $O/gputests_pin3.txt 2>&1; tail -2 $O/gputests_pin3.txt
for r in 1; do
for pin in 0 1 3; do
  echo "== ISSCABAC_PAIR_PIN=$pin"; ISSCABAC_PAIR_PIN=$pin python tools/s_sweep.py --streams 1048576,131072,65536,32768,16384 2>&1 | grep -v amdgpu
  echo "-- config 5's shape (--ep --bins 8192)"; ISSCABAC_PAIR_PIN=$pin python tools/s_sweep.py --ep --bins 8192 --streams 1048576,131072,65536 2>&1 | grep -v amdgpu
done; done > $O/sweep_by_pin.txt 2>&1; cat $O/sweep_by_pin.txt
