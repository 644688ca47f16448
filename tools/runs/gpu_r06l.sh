#!/bin/bash
# round 6, run l: k_gather with two steps' loads in flight -- gather tests, A/B on config 3 and config 5's shape
O=gpurun_out/r06l; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py -x -q -m gpu > $O/gputests.txt 2>&1; tail -3 $O/gputests.txt
python tools/ab_bench.py tools/ab/variant_base.so isscabac_amd/lib/libisscabac_hip.so --rounds 3 > $O/ab_gather_config3.txt 2>&1; cat $O/ab_gather_config3.txt
python tools/ab_bench.py tools/ab/variant_base.so isscabac_amd/lib/libisscabac_hip.so --rounds 2 -- --workload config5_per_gpu > $O/ab_gather_config5.txt 2>&1; cat $O/ab_gather_config5.txt
