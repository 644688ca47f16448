#!/bin/bash
# round 6, run af: workgroup sizes with every size placed evenly (the most waves; 2 > 4 > 1 > 3 among equals) -- parity, context-count rates, soaks
O=gpurun_out/r06af; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/gputests.txt 2>&1; tail -3 $O/gputests.txt
for r in 1 2 3; do python tools/ctx_rates.py 2>&1 | grep -v amdgpu; done > $O/ctx_rates.txt; cat $O/ctx_rates.txt
timeout 200 python tests/soak/gpu_fuzz.py 120 681 > $O/soak.txt 2>&1; tail -1 $O/soak.txt
timeout 200 python tests/soak/gpu_fuzz_bucketed.py 90 682 > $O/soak_bucketed.txt 2>&1; tail -1 $O/soak_bucketed.txt
