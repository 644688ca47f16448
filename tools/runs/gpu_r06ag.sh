#!/bin/bash
# round 6, run ag: k_encode's tables are the 1 KiB of entries alone (the lane-by-lane step on them) -- parity, context-count rates, config 3 / 5 A/B vs HEAD, soaks
O=gpurun_out/r06ag; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/gputests.txt 2>&1; tail -3 $O/gputests.txt
for r in 1 2 3; do python tools/ctx_rates.py 2>&1 | grep -v amdgpu; done > $O/ctx_rates.txt; cat $O/ctx_rates.txt
python tools/ab_bench.py tools/ab/variant_base.so isscabac_amd/lib/libisscabac_hip.so --rounds 2 > $O/ab_config3.txt 2>&1; cat $O/ab_config3.txt
python tools/ragged_rate.py 2>&1 | grep -v amdgpu > $O/ragged.txt; cat $O/ragged.txt
timeout 300 python tests/soak/gpu_fuzz.py 150 691 > $O/soak.txt 2>&1; tail -1 $O/soak.txt
timeout 200 python tests/soak/gpu_fuzz_bucketed.py 90 692 > $O/soak_bucketed.txt 2>&1; tail -1 $O/soak_bucketed.txt
