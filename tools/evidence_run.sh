# usage (GPU box, repo root): bash tools/evidence_run.sh <outdir>
# bench line, rocprof kernel stats, PMC passes, the stream-count sweep and the other-shape rates of the current build into
# <outdir> (under gpurun_out/); then, back in the build container: python tools/refresh_profiles.py <outdir> r06
set -u
O=${1:-gpurun_out/r06z}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --steps 5 --warmup 1 --cpu-seconds 0 --no-verify > $O/stats.log 2>&1
PMC_MEM=1 bash tools/pmc_case.sh $O/pmc_config3 > /dev/null 2>&1
python3 tools/pmc_summary.py $O/pmc_config3 > $O/pmc_summary.json
# the bench line quotes this build's counter traffic: profiles/pmc_traffic.json first, then bench.py
python3 tools/refresh_profiles.py --traffic-only $O/pmc_summary.json
python bench.py > $O/bench.json 2> $O/bench.err
python3 tools/pmc_case_table.py $O/pmc_config3 1048576 4096 > $O/pmc_table.txt
PMC_MEM=1 bash tools/pmc_case.sh $O/pmc_config5 --workload config5_per_gpu > /dev/null 2>&1
python3 tools/pmc_case_table.py $O/pmc_config5 1048576 8192 >> $O/pmc_table.txt
bash tools/pmc_case.sh $O/pmc_128k --streams 131072 > /dev/null 2>&1
python3 tools/pmc_case_table.py $O/pmc_128k 131072 4096 >> $O/pmc_table.txt
bash tools/pmc_case.sh $O/pmc_16k --streams 16384 > /dev/null 2>&1
python3 tools/pmc_case_table.py $O/pmc_16k 16384 4096 >> $O/pmc_table.txt
python tools/s_sweep.py --out $O/s_sweep.json > $O/s_sweep.txt 2>&1
{
echo "== bench.py --workload config5 (8 Mi streams x 8 Ki bins on ONE GPU, 8 launches per step)"; python bench.py --workload config5 --steps 3 --warmup 1 --cpu-seconds 0 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['kernels_ms'], d['config']['workload'], 'verified', d['verified_bit_exact'])"
echo "== bench.py --workload config5_per_gpu"; python bench.py --workload config5_per_gpu --steps 5 --cpu-seconds 0 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['kernels_ms'], d['config']['workload'])"
echo "== bench.py --workload config2"; python bench.py --workload config2 --steps 50 --warmup 3 --cpu-seconds 0 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['kernels_ms'], d['config']['workload'])"
echo "== tools/ctx_rates.py"; python tools/ctx_rates.py
echo "== tools/ragged_rate.py"; python tools/ragged_rate.py
echo "== tools/symbol_rate.py"; python tools/symbol_rate.py
echo "== tools/host_api_rate.py"; python tools/host_api_rate.py
echo "== tools/per_bin_rate.py"; python tools/per_bin_rate.py
echo "== tests/perf/per_bin_native.cpp (the per-bin commands from C++; second run: ISSCABAC_STREAM_LIVE=0)"
mkdir -p tests/_build && g++ -O2 -std=c++17 -I include tests/perf/per_bin_native.cpp -o tests/_build/per_bin_native -L isscabac_amd/lib -lisscabac_hip -Wl,-rpath,"$GRAFT_REPO_ROOT/isscabac_amd/lib" -Wl,--allow-shlib-undefined && { LD_LIBRARY_PATH=/opt/rocm/lib:${LD_LIBRARY_PATH:-} timeout 300 tests/_build/per_bin_native; LD_LIBRARY_PATH=/opt/rocm/lib:${LD_LIBRARY_PATH:-} ISSCABAC_STREAM_LIVE=0 timeout 300 tests/_build/per_bin_native; }
echo "== tools/wide_rate.py"; python tools/wide_rate.py 65536 1024 999; python tools/wide_rate.py 1048576 1024 999; python tools/wide_rate.py 1048576 1024 500; python tools/wide_rate.py 1048576 1024 256
echo "== tools/init_rate.py"; python tools/init_rate.py
echo "== tools/iss_matrix_rate.py"; python tools/iss_matrix_rate.py
echo "== tools/live_interleave.py"; python tools/live_interleave.py
echo "== tests/perf/symbol_host_rate.py --real-only"; python tests/perf/symbol_host_rate.py --real-only
} > $O/other_shapes.txt 2>&1
tail -3 $O/other_shapes.txt
