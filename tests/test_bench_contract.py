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
    assert d["gather_ok"] is None and d["gather_ms"] is None and d["gather_GBps"] is None and d["dist"] is None  # one rank: nothing to gather
    assert len(d["per_rank"]) == 1 and d["per_rank"][0]["encode_instance"] == "k_encode_pair"
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
    # encode-only and decode-only rates beside the round trip, from the engine's HIP events
    k = d["kernels_ms"]
    bins_per_launch = d["config"]["streams_per_gpu"] * d["config"]["bins_per_stream"] / d["config"]["launches_per_step"]
    assert abs(d["encode_Gbin_s"] - bins_per_launch / ((k["encode"] + k["scan"] + k["gather"]) * 1e-3) / 1e9) < 1e-6 * d["encode_Gbin_s"]
    assert abs(d["decode_Gbin_s"] - bins_per_launch / (k["decode"] * 1e-3) / 1e9) < 1e-6 * d["decode_Gbin_s"]
    assert d["encode_Gbin_s"] > d["value"] and d["decode_Gbin_s"] > d["value"] and "decode" in d["step"]
    # value = bins / wall time, consistent with ms_per_step
    assert d["config"]["streams_total"] == 8192 and d["config"]["launches_per_step"] == 1
    bins_per_step = d["config"]["streams_total"] * d["config"]["bins_per_stream"]
    assert abs(d["value"] - bins_per_step / (d["ms_per_step"] * 1e-3) / 1e9) < 1e-6 * d["value"]


@pytest.mark.gpu
def test_bench_config2_is_encode_only_as_baseline_names_it():
    """BASELINE.json config 2: "64K streams x 1K bins, 3 contexts, encode-only".  The default step of --workload config2 is
    encode (kernel + scan + gather) alone; --round-trip times the way back too.  Either way the line carries both rates and
    the payload has been decoded back and compared once, outside the timed region."""
    def run(*extra):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "config2", "--steps", "3", "--warmup", "1",
                              "--cpu-seconds", "1"] + list(extra), capture_output=True, text=True, cwd=ROOT, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        return json.loads(out.stdout.strip().splitlines()[-1])
    d = run()
    assert "encode-only" in d["metric"] and "65536 streams x 1024 bins, 3 contexts" in d["metric"]
    assert d["step"] == "encode (kernel + scan + gather)" and d["verified_bit_exact"] is True
    assert d["roofline"]["kernel"] == "k_encode" and d["encode_Gbin_s"] > 0 and d["decode_Gbin_s"] > 0
    bins = d["config"]["streams_total"] * d["config"]["bins_per_stream"]
    assert abs(d["value"] - bins / (d["ms_per_step"] * 1e-3) / 1e9) < 1e-6 * d["value"]
    assert d["value"] <= d["encode_Gbin_s"] * 1.05  # the timed step is the encode path alone
    r = run("--round-trip")
    assert "encode+decode round-trip" in r["metric"] and "decode" in r["step"] and r["value"] < d["value"]


def _run_dist_bench(extra_env, tag):
    """bench.py under the launcher the driver uses for N > 1, with one rank: ISSCABAC_FORCE_DIST=1 takes the multi-rank
    branch (RCCL process group, barriers, all_reduce of the elapsed time, the bitstream gather).  The launcher starts
    before anything touches the GPU; nothing is re-executed afterwards."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, ISSCABAC_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", **extra_env)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
           "--streams", "8192", "--cpu-seconds", "0"]
    out = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT, timeout=600, env=env)
    logdir = os.path.join(ROOT, "gpurun_out", "r04_dist")
    try:  # kept as evidence (copied to profiles/r04/ by hand): the RCCL branch of bench.py ran on this build
        os.makedirs(logdir, exist_ok=True)
        with open(os.path.join(logdir, tag + ".log"), "w") as f:
            f.write("$ %s\n[exit %d]\n--- stdout\n%s\n--- stderr (tail)\n%s\n" % (" ".join(cmd), out.returncode, out.stdout, out.stderr[-3000:]))
    except OSError:
        pass
    return out


@pytest.mark.gpu
def test_bench_multi_rank_branch_runs_with_one_rank():
    out = _run_dist_bench({}, "force_dist_ok")
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["gather_ok"] is True and d["gather_ms"] > 0 and d["gather_bytes"] == d["config"]["payload_bytes_per_gpu"]
    assert d["n_gpus"] == 1 and d["verified_bit_exact"] is True and d["value"] > 0
    assert d["step_kernel_ms_min"] <= d["step_kernel_ms_median"]
    # a first N-GPU run explains itself: the group as RCCL reports it, the gather's rate, every rank's time, shard and kernels
    assert d["dist"] == {"backend": "nccl", "world_size": 1, "launcher_world_size": 1}
    assert abs(d["gather_GBps"] - d["gather_bytes"] / (d["gather_ms"] * 1e-3) / 1e9) < 1e-6 * d["gather_GBps"]
    assert len(d["per_rank"]) == 1
    r0 = d["per_rank"][0]
    assert r0["rank"] == 0 and r0["streams"] == 8192 and r0["payload_bytes"] == d["config"]["payload_bytes_per_gpu"]
    assert 0 < r0["ms_per_step"] <= d["ms_per_step"] * 1.0001  # the line's figure is the MAX over ranks
    assert r0["encode_instance"] == "k_encode_pair" and r0["decode_instance"] == "k_decode_pair"  # 8 192 streams under AUTO
    assert set(r0["kernels_ms"]) == {"encode", "scan", "gather", "decode"}


@pytest.mark.gpu
def test_bench_gather_deadline_fails_visibly_after_the_line():
    out = _run_dist_bench({"ISSCABAC_GATHER_DEADLINE_S": "0"}, "force_dist_deadline")
    lines = [l for l in out.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, out.stdout  # the result line is out ...
    d = json.loads(lines[0])
    assert d["gather_ok"] is False and d["value"] > 0
    assert out.returncode != 0  # ... and the run failed: the launcher's failure record names the rank's exit code 3
    import re
    assert re.search(r"exitcode\s*:\s*3\b", out.stderr), out.stderr[-1500:]


@pytest.mark.gpu
def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus N` with no launcher around it (the form the driver uses for N = 1) must start the ranks
    itself as child processes.  One GPU here, so: one rank, forced through the launcher (ISSCABAC_SELF_LAUNCH) and through
    the multi-rank branch (ISSCABAC_FORCE_DIST) -- RCCL group, barriers, MAX reduction, gather into the root's buffer."""
    env = dict(os.environ, ISSCABAC_FORCE_DIST="1", ISSCABAC_SELF_LAUNCH="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--streams", "8192",
           "--cpu-seconds", "0"]
    out = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT, timeout=600, env=env)
    logdir = os.path.join(ROOT, "gpurun_out", "r04_dist")
    try:
        os.makedirs(logdir, exist_ok=True)
        with open(os.path.join(logdir, "self_launch.log"), "w") as f:
            f.write("$ ISSCABAC_FORCE_DIST=1 ISSCABAC_SELF_LAUNCH=1 %s\n[exit %d]\n--- stdout\n%s\n--- stderr (tail)\n%s\n"
                    % (" ".join(cmd), out.returncode, out.stdout, out.stderr[-3000:]))
    except OSError:
        pass
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["gather_ok"] is True and d["gather_bytes"] == d["config"]["payload_bytes_per_gpu"] and d["verified_bit_exact"] is True


def test_bench_self_launch_two_ranks_meet_without_a_gpu():
    """the launcher half of `python bench.py --gpus 2` on CPU: two child ranks under torch.distributed.run rendezvous on
    127.0.0.1 (gloo), reduce, rank 0 prints one line, the parent returns the launcher's status"""
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rendezvous-only"],
                         capture_output=True, text=True, cwd=ROOT, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0]) == {"rendezvous_only": True, "world": 2, "max_rank_plus_1": 2.0}
    # a rank that fails makes the whole call fail: --gpus disagreeing with the world the launcher made
    bad = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
                          "127.0.0.1", "--master-port", "0", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rendezvous-only"],
                         capture_output=True, text=True, cwd=ROOT, timeout=300, env=env)
    assert bad.returncode != 0 and "WORLD_SIZE=1" in bad.stderr


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
