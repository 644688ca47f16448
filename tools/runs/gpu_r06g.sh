#!/bin/bash
# round 6, run g: k_decode on PieceSrc after the address fix -- full suite, soaks, context-count rates
O=gpurun_out/r06g; mkdir -p $O
timeout 1200 python -m pytest tests -x -q -m gpu > $O/gputests.txt 2>&1; tail -3 $O/gputests.txt
timeout 300 python tests/soak/gpu_fuzz.py 120 64 > $O/soak.txt 2>&1; tail -1 $O/soak.txt
timeout 300 python tests/soak/gpu_fuzz_bucketed.py 45 65 > $O/soak_bucketed.txt 2>&1; tail -1 $O/soak_bucketed.txt
python tools/ctx_rates.py > $O/ctx_rates.txt 2>&1; cat $O/ctx_rates.txt
python tools/ab_bench.py tools/ab/variant_base.so isscabac_amd/lib/libisscabac_hip.so --rounds 3 -- --streams 262144 > $O/ab_pieces_256k.txt 2>&1; cat $O/ab_pieces_256k.txt
