# usage (GPU box, repo root): bash tools/evidence_run.sh -- bench line, rocprof kernel stats, PMC passes and the other-shape
# rates of the current build into gpurun_out/r01h; then: python tools/refresh_profiles.py gpurun_out/r01h r01
set -u
O=gpurun_out/r01h; mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --steps 5 --warmup 1 --cpu-seconds 0 --no-verify > $O/stats.log 2>&1
bash tools/pmc_quick.sh $O/pmc > /dev/null 2>&1
python3 tools/pmc_summary.py $O/pmc > $O/pmc_summary.json
python3 tools/pmc_table.py $O/pmc > $O/pmc_table.txt
{
echo "== bench.py --workload config5_per_gpu"; python bench.py --workload config5_per_gpu --steps 3 --cpu-seconds 0 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['kernels_ms'], d['config']['workload'])"
echo "== bench.py --workload config2"; python bench.py --workload config2 --steps 20 --warmup 3 --cpu-seconds 0 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['kernels_ms'], d['config']['workload'])"
echo "== tools/ctx_rates.py"; python tools/ctx_rates.py
echo "== tools/ragged_rate.py"; python tools/ragged_rate.py
echo "== tools/symbol_rate.py"; python tools/symbol_rate.py
echo "== tools/host_api_rate.py"; python tools/host_api_rate.py
} > $O/other_shapes.txt 2>&1
tail -3 $O/other_shapes.txt
