#!/bin/bash
# round 6, run i: k_encode with 16 bytes per stream resident for more than 64 contexts (1 KiB tiles) -- parity, context-count rates A/B
O=gpurun_out/r06i; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/gputests.txt 2>&1; tail -3 $O/gputests.txt
for r in 1 2; do
  echo "== base (32 bytes resident)"; ISSCABAC_LIB=$PWD/tools/ab/variant_base.so python tools/ctx_rates.py
  echo "== 16 bytes resident";        python tools/ctx_rates.py
done > $O/ab_ctx_rates.txt 2>&1; cat $O/ab_ctx_rates.txt
timeout 200 python tests/soak/gpu_fuzz.py 60 67 > $O/soak.txt 2>&1; tail -1 $O/soak.txt
timeout 200 python tests/soak/gpu_fuzz_bucketed.py 45 68 > $O/soak_bucketed.txt 2>&1; tail -1 $O/soak_bucketed.txt
