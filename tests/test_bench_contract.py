"""bench.py prints exactly one JSON line with the fields the driver and the judge read (prompt section 4).  Run on a small
number of streams so the test takes seconds; the headline run is `python bench.py` with no flags."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_json_line_contract():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                          "--streams", "8192", "--cpu-seconds", "2"], capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for key, typ in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                     ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str),
                     ("config", dict), ("roofline", dict), ("cpu_baseline", dict)):
        assert isinstance(d[key], typ), (key, d[key])
    # the BASELINE.json metric string is only true for its own shape: a reduced run names the shape it really coded
    assert d["metric"] != json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"] and "8192 streams x 4096 bins" in d["metric"]
    assert d["gather_ok"] is None and d["gather_ms"] is None  # one rank: nothing to gather
    assert d["unit"] == "Gbin/s" and d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1
    assert d["scaling"] == "strong" and d["vs_baseline"] is None and d["data"] == "synthetic" and d["higher_is_better"] is True
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0 < r["frac"] < 1
    assert r["traffic"] is None  # PMC traffic is per launch of the full workload only
    c = d["cpu_baseline"]
    assert c["unit"] == "Gbin/s" and c["cores"] == 1 and c["kind"] in ("reference", "port") and c["value"] > 0 and c["sample"]
    assert c["gpu_streams_checked_byte_exact"] == 4096
    assert d["verified_bit_exact"] is True
    # value = bins / wall time, consistent with ms_per_step
    assert d["config"]["streams_total"] == 8192 and d["config"]["launches_per_step"] == 1
    bins_per_step = d["config"]["streams_total"] * d["config"]["bins_per_stream"]
    assert abs(d["value"] - bins_per_step / (d["ms_per_step"] * 1e-3) / 1e9) < 1e-6 * d["value"]


def test_default_metric_string_is_baseline_json_verbatim():
    sys.path.insert(0, ROOT)
    import bench
    assert bench.BASELINE_METRIC == json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    w = bench.WORKLOADS
    assert (w["config3"]["S"], w["config3"]["B"], w["config3"]["C"]) == (1 << 20, 4096, 64)
    assert (w["config5"]["S"], w["config5"]["B"], w["config5"]["ep"]) == (8 << 20, 8192, True)
    assert (w["config2"]["S"], w["config2"]["B"], w["config2"]["C"]) == (1 << 16, 1024, 3)


def test_graft_entry_points_exist():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    assert callable(g.build) and callable(g.smoke)
