#!/bin/bash
# round 6, run ah: the <= 64 context kernels in pinned two-wave workgroups (-DISSCABAC_ENC_WPB=2 -DISSCABAC_DEC_WPB=2) against four-wave ones
O=gpurun_out/r06ah; mkdir -p $O
ISSCABAC_LIB=$PWD/tools/ab/variant_wpb2.so timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "throughput and not native_library" > $O/gputests_wpb2.txt 2>&1; tail -2 $O/gputests_wpb2.txt
python tools/ab_bench.py isscabac_amd/lib/libisscabac_hip.so tools/ab/variant_wpb2.so --rounds 3 > $O/ab_config3.txt 2>&1; cat $O/ab_config3.txt
python tools/ab_bench.py isscabac_amd/lib/libisscabac_hip.so tools/ab/variant_wpb2.so --rounds 2 -- --workload config5_per_gpu > $O/ab_config5.txt 2>&1; cat $O/ab_config5.txt
python tools/ab_bench.py isscabac_amd/lib/libisscabac_hip.so tools/ab/variant_wpb2.so --rounds 2 -- --streams 262144 > $O/ab_256k.txt 2>&1; cat $O/ab_256k.txt
