#!/bin/bash
# round 6, run q: k_encode sends the next tile part to LDS behind the current part's last reads -- parity, A/B on config 3 / config 5 / context counts
O=gpurun_out/r06q; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/gputests.txt 2>&1; tail -3 $O/gputests.txt
python tools/ab_bench.py tools/ab/variant_base.so isscabac_amd/lib/libisscabac_hip.so --rounds 3 > $O/ab_early_config3.txt 2>&1; cat $O/ab_early_config3.txt
python tools/ab_bench.py tools/ab/variant_base.so isscabac_amd/lib/libisscabac_hip.so --rounds 2 -- --workload config5_per_gpu > $O/ab_early_config5.txt 2>&1; cat $O/ab_early_config5.txt
for r in 1 2; do
  echo "== base"; ISSCABAC_LIB=$PWD/tools/ab/variant_base.so python tools/ctx_rates.py
  echo "== next part stored early";        python tools/ctx_rates.py
done > $O/ab_ctx_rates.txt 2>&1; grep -v amdgpu $O/ab_ctx_rates.txt
timeout 200 python tests/soak/gpu_fuzz.py 90 631 > $O/soak.txt 2>&1; tail -1 $O/soak.txt
