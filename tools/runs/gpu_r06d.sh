#!/bin/bash
# round 6, run d: decoder mixed step on a dummy context (dec_bin_mixed2_nx_abs) -- parity, config 5 A/B against the previous build
O=gpurun_out/r06d; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_f4.py -x -q -m gpu > $O/parity.txt 2>&1; tail -3 $O/parity.txt
timeout 600 python -m pytest tests/test_gpu_scale.py -x -q -m gpu -k "config5_per_gpu or per_gpu_share" > $O/parity_scale.txt 2>&1; tail -3 $O/parity_scale.txt
timeout 300 python tests/soak/gpu_fuzz.py 90 61 > $O/soak.txt 2>&1; tail -1 $O/soak.txt
python tools/ab_bench.py tools/ab/variant_line.so isscabac_amd/lib/libisscabac_hip.so --rounds 4 --workload config5_per_gpu > $O/ab_config5.txt 2>&1; cat $O/ab_config5.txt
python tools/ab_bench.py tools/ab/variant_line.so isscabac_amd/lib/libisscabac_hip.so --rounds 4 > $O/ab_config3.txt 2>&1; cat $O/ab_config3.txt
PMC_MEM=0 bash tools/pmc_case.sh $O/pmc_config5 --workload config5_per_gpu > /dev/null 2>&1
python3 tools/pmc_case_table.py $O/pmc_config5 1048576 8192 > $O/pmc_table.txt; cat $O/pmc_table.txt
