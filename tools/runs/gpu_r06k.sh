#!/bin/bash
# round 6, run k: workgroup sizes by the CU's real LDS granule (1280 bytes) -- k_encode with more than 64 contexts in 2-wave workgroups
O=gpurun_out/r06k; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/gputests.txt 2>&1; tail -3 $O/gputests.txt
for r in 1 2; do
  echo "== base (single-wave encoder workgroups above 64 contexts)"; ISSCABAC_LIB=$PWD/tools/ab/variant_base.so python tools/ctx_rates.py
  echo "== workgroup size by granules";        python tools/ctx_rates.py
done > $O/ab_ctx_rates.txt 2>&1; cat $O/ab_ctx_rates.txt
