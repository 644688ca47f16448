#!/bin/bash
# round 6, run ac: the encoder's latency instance for every launch of more than 64 contexts -- parity, context-count rates
O=gpurun_out/r06ac; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/gputests.txt 2>&1; tail -3 $O/gputests.txt
for r in 1 2; do python tools/ctx_rates.py 2>&1 | grep -v amdgpu; done > $O/ctx_rates.txt; cat $O/ctx_rates.txt
timeout 200 python tests/soak/gpu_fuzz.py 100 671 > $O/soak.txt 2>&1; tail -1 $O/soak.txt
