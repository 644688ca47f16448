#!/bin/bash
# round 4: 128-byte segments for both encoder arrays on the final build: time (config 3, config 5, 128 Ki streams) and fills
set -u
OUT=gpurun_out/r04p; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ISSCABAC_LIB=$PWD/tools/ab/variant_lpr88.so timeout 300 python3 bench.py --steps 2 --warmup 1 --cpu-seconds 4 > $OUT/verify_lpr88.json 2> $OUT/verify_lpr88.err; echo "lpr88 verify rc=$?"
timeout 900 python3 tools/ab_bench.py tools/ab/variant_cur.so tools/ab/variant_lpr88.so --rounds 8 > $OUT/ab_config3.txt 2>&1; cat $OUT/ab_config3.txt
timeout 900 python3 tools/ab_bench.py tools/ab/variant_cur.so tools/ab/variant_lpr88.so --rounds 4 --workload config5_per_gpu > $OUT/ab_config5.txt 2>&1; cat $OUT/ab_config5.txt
for l in cur lpr88; do ISSCABAC_LIB=$PWD/tools/ab/variant_$l.so timeout 300 python3 tools/s_sweep.py --streams 131072,65536 > $OUT/sweep_$l.txt 2>&1; grep streams $OUT/sweep_$l.txt; done
bash tools/pmc_fetch.sh $OUT/pmc tools/ab/variant_lpr88.so > $OUT/fetch.txt 2>&1; cat $OUT/fetch.txt
