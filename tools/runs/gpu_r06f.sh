#!/bin/bash
# round 6, run f: k_decode on aligned, non-overlapping payload pieces (PieceSrc) -- parity, soak, same-box A/B, FETCH
O=gpurun_out/r06f; mkdir -p $O
timeout 1200 python -m pytest tests -x -q -m gpu > $O/gputests.txt 2>&1; tail -3 $O/gputests.txt
timeout 300 python tests/soak/gpu_fuzz.py 90 64 > $O/soak.txt 2>&1; tail -1 $O/soak.txt
timeout 300 python tests/soak/gpu_fuzz_bucketed.py 45 65 > $O/soak_bucketed.txt 2>&1; tail -1 $O/soak_bucketed.txt
python tools/ab_bench.py tools/ab/variant_base.so isscabac_amd/lib/libisscabac_hip.so --rounds 5 > $O/ab_pieces.txt 2>&1; cat $O/ab_pieces.txt
python tools/ab_bench.py tools/ab/variant_base.so isscabac_amd/lib/libisscabac_hip.so --rounds 3 --workload config5_per_gpu > $O/ab_pieces_c5.txt 2>&1; cat $O/ab_pieces_c5.txt
python tools/ab_bench.py tools/ab/variant_base.so isscabac_amd/lib/libisscabac_hip.so --rounds 3 -- --streams 131072 > $O/ab_pieces_128k.txt 2>&1; cat $O/ab_pieces_128k.txt
bash tools/pmc_fetch.sh $O/pmc isscabac_amd/lib/libisscabac_hip.so > $O/fetch.txt 2>&1; cat $O/fetch.txt
python tools/ctx_rates.py > $O/ctx_rates.txt 2>&1; cat $O/ctx_rates.txt
