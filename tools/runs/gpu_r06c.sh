#!/bin/bash
# round 6, run c: whole-line encoder tiles as the default + host twins -- full GPU suite, bench, other shapes
O=gpurun_out/r06c; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/gputests.txt 2>&1; echo "pytest rc $?" >> $O/gputests.txt; tail -4 $O/gputests.txt
python bench.py > $O/bench.json 2> $O/bench.err; python -c "
import json; d=json.load(open('$O/bench.json')); print(d['value'], d['kernels_ms'], d['roofline']['frac'])"
{
echo "== tools/ctx_rates.py"; python tools/ctx_rates.py
echo "== bench.py --workload config5_per_gpu"; python bench.py --workload config5_per_gpu --steps 5 --cpu-seconds 0 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['kernels_ms'], d['config']['workload'])"
echo "== bench.py --workload config2"; python bench.py --workload config2 --steps 50 --warmup 3 --cpu-seconds 0 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['kernels_ms'], d['config']['workload'])"
echo "== tools/ragged_rate.py"; python tools/ragged_rate.py
echo "== tools/iss_matrix_rate.py"; python tools/iss_matrix_rate.py
echo "== tests/perf/per_bin_native.cpp"
g++ -O2 -std=c++17 -I include tests/perf/per_bin_native.cpp -o tests/_build/per_bin_native -L isscabac_amd/lib -lisscabac_hip -Wl,-rpath,"$GRAFT_REPO_ROOT/isscabac_amd/lib" -Wl,--allow-shlib-undefined && LD_LIBRARY_PATH=/opt/rocm/lib:${LD_LIBRARY_PATH:-} timeout 300 tests/_build/per_bin_native
} > $O/other_shapes.txt 2>&1
cat $O/other_shapes.txt
python tools/s_sweep.py --out $O/s_sweep.json > $O/s_sweep.txt 2>&1; cat $O/s_sweep.txt
