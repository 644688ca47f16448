#!/bin/bash
# round 4: the encoder's mixed step on a dummy context (config 5): GPU suite, A/B on config 5's per-GPU shape and on config 3, soak
set -u
OUT=gpurun_out/r04m; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 1200 python3 -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.txt 2>&1
tail -4 $OUT/pytest_gpu.txt
timeout 1500 python3 tools/ab_bench.py tools/ab/variant_mixed1.so tools/ab/variant_mixed2.so --rounds 5 --workload config5_per_gpu > $OUT/ab_config5.txt 2>&1
cat $OUT/ab_config5.txt
timeout 900 python3 tools/ab_bench.py tools/ab/variant_mixed1.so tools/ab/variant_mixed2.so --rounds 4 > $OUT/ab_config3.txt 2>&1
cat $OUT/ab_config3.txt
timeout 300 python3 tests/soak/gpu_fuzz.py 150 501 > $OUT/soak_bins.txt 2>&1; tail -1 $OUT/soak_bins.txt
timeout 300 python3 tests/soak/gpu_fuzz_bucketed.py 100 502 > $OUT/soak_bucketed.txt 2>&1; tail -1 $OUT/soak_bucketed.txt
