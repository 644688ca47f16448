#!/bin/bash
# round 6, run aa: the decoder pair pinned and used up to 128 Ki streams -- full parity, sweeps A/B (config 3's and config 5's shapes, config 2), soaks under AUTO
O=gpurun_out/r06aa; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/gputests.txt 2>&1; tail -3 $O/gputests.txt
for r in 1 2; do for lib in tools/ab/variant_base.so isscabac_amd/lib/libisscabac_hip.so; do
  echo "== $lib"; ISSCABAC_LIB=$PWD/$lib python tools/s_sweep.py --streams 1048576,262144,131072,65536,32768 2>&1 | grep -v amdgpu
  echo "-- config 5's shape"; ISSCABAC_LIB=$PWD/$lib python tools/s_sweep.py --ep --bins 8192 --streams 1048576,131072,65536 2>&1 | grep -v amdgpu
  echo "-- config 2 round trip"; ISSCABAC_LIB=$PWD/$lib python bench.py --workload config2 --round-trip --steps 50 --warmup 3 --cpu-seconds 0 --no-verify 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['kernels_ms'])"
done; done > $O/ab_sweep.txt 2>&1; cat $O/ab_sweep.txt
timeout 300 python tests/soak/gpu_fuzz.py 150 661 > $O/soak.txt 2>&1; tail -1 $O/soak.txt
