#!/bin/bash
# round 6, run h: k_encode's mixed step on a dummy context that holds the bypass state (enc_bin_mixed3_acc) -- parity, config 5 A/B
O=gpurun_out/r06h; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/gputests.txt 2>&1; tail -3 $O/gputests.txt
python tools/ab_bench.py tools/ab/variant_base.so isscabac_amd/lib/libisscabac_hip.so --rounds 3 -- --workload config5_per_gpu > $O/ab_mixed3_config5.txt 2>&1; cat $O/ab_mixed3_config5.txt
python tools/ab_bench.py tools/ab/variant_base.so isscabac_amd/lib/libisscabac_hip.so --rounds 2 > $O/ab_mixed3_config3.txt 2>&1; cat $O/ab_mixed3_config3.txt
timeout 300 python tests/soak/gpu_fuzz.py 90 66 > $O/soak.txt 2>&1; tail -1 $O/soak.txt
