#!/bin/bash
# round 5, call a: baseline of the round-4 build on this box (tests, bench line, stream-count sweep) and the
# L2 <-> fabric request counters at a launch size the TCC passes survive
set -u
OUT=gpurun_out/r05a; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 400 python3 -m pytest tests -x -q -m gpu > $OUT/gpu_tests.txt 2>&1; echo "pytest rc=$?" | tee -a $OUT/rc.txt
timeout 300 python3 bench.py --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" | tee -a $OUT/rc.txt
timeout 300 python3 tools/s_sweep.py --out $OUT/s_sweep.json > $OUT/s_sweep.txt 2>&1; echo "sweep rc=$?" | tee -a $OUT/rc.txt
rocprofv3 -L > $OUT/counters_list.txt 2>&1
run() { name=$1; S=$2; shift; shift
  timeout 120 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 bench.py --steps 2 --warmup 1 --cpu-seconds 0 --no-verify --streams $S --tuning throughput > "$OUT/$name.log" 2>&1
  echo "$name rc=$?" | tee -a $OUT/rc.txt; }
for S in 262144; do
  run ea_rd_$S $S TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
  run ea_rd_dram_$S $S TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_DRAM_sum
  run ea_wr_$S $S TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
  run hitmiss_$S $S TCC_HIT_sum TCC_MISS_sum
  run req_$S $S TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum
  run fetch_$S $S FETCH_SIZE
  run write_$S $S WRITE_SIZE
done
python3 tools/pmc_summary.py $OUT > $OUT/pmc_summary.json 2>$OUT/pmc_summary.err
tail -8 $OUT/s_sweep.txt; cat $OUT/rc.txt
