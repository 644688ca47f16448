#!/bin/bash
# round 6, run e: workgroup size by LDS fit for > 64 contexts; the decoder's payload traffic priced (timing-only patch)
O=gpurun_out/r06e; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_f4.py tests/test_gpu_wide.py -x -q -m gpu > $O/parity.txt 2>&1; tail -3 $O/parity.txt
timeout 300 python tests/soak/gpu_fuzz.py 60 62 > $O/soak.txt 2>&1; tail -1 $O/soak.txt
timeout 300 python tests/soak/gpu_fuzz_bucketed.py 45 63 > $O/soak_bucketed.txt 2>&1; tail -1 $O/soak_bucketed.txt
python tools/ctx_rates.py > $O/ctx_rates.txt 2>&1; cat $O/ctx_rates.txt
python tools/ab_bench.py tools/ab/variant_base.so tools/ab/variant_nopayload.so tools/ab/variant_nostore.so tools/ab/variant_noload.so --rounds 4 > $O/ab_memory_side.txt 2>&1; cat $O/ab_memory_side.txt
bash tools/pmc_fetch.sh $O/pmc tools/ab/variant_base.so tools/ab/variant_nopayload.so > $O/fetch.txt 2>&1; cat $O/fetch.txt
