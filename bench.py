#!/usr/bin/env python3
"""bench.py -- encode+decode round-trip throughput of the batched CABAC engine on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload config3|config5|config2] [--scaling strong|weak]

One "step" = one pass of the hot path over the workload: encode (kernel + offsets scan + payload gather) followed by
decode, inputs already resident in HBM.  The steps are enqueued through the engine's asynchronous entry points
(isscabac_*_batch_dev_async): the host reads nothing between kernels, so the figure does not depend on host wake-up
latency; sizes, the capacity check and the kernel times are read after the closing fence.

Workloads are BASELINE.json's configurations, named by their GLOBAL shape:
    config3   1 Mi streams x 4 Ki bins, 64 contexts                        (the shape `metric` is quoted on; default)
    config5   8 Mi streams x 8 Ki bins, 64 contexts, 50 % bypass bins      (one GPU runs it 1 Mi streams at a time)
    config2   64 Ki streams x 1 Ki bins, 3 contexts, ENCODE-ONLY as BASELINE.json names it (a step = encode kernel +
              offsets scan + payload gather; --round-trip times encode + decode like the others).  Every line also
              carries encode_Gbin_s (kernel + scan + gather) and decode_Gbin_s from the engine's HIP events (SURVEY 8(d):
              "report encode-only and decode-only too").
Streams are independent, so N ranks split the stream range and nothing is exchanged while coding:
    --scaling strong   the global shape is fixed, rank r codes sharding.shard_range(S, r, N) -- the mode under which the
                       metric's "1M streams x 4K bins" stays true for every N, and the north star's efficiency figure;
                       the default
    --scaling weak     every rank codes the workload's per-GPU shape (config3: 1 Mi streams; config5: 1 Mi; config2:
                       64 Ki); the job grows with N
`value` = bins coded by all ranks / wall time between two fences (barrier + device synchronise), MAX over ranks.  The
final bitstream gather to rank 0 (the north star's only exchange) is timed apart (gather_ms), outside `value`; if it does
not complete the line says "gather_ok": false and the process exits non-zero after the line is flushed.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (the longer of the encode / decode kernels), its
duration measured with HIP events on the launch stream inside the engine; algorithmic bytes per launch follow SURVEY.md
8(d): encode reads 2 B/bin and writes P + 8 B/stream, decode reads 1 B/bin + P + 8 B/stream and writes 1 B/bin (P =
measured payload bytes of one launch).  `cpu_baseline` (rank 0) times the real reference classes (oracle/_ref, kind
"reference") or, where that build is absent, the C restatement (kind "port") on a bounded sample of the same workload on
the host cores of this box, and -- the only place bench.py touches oracle/ -- checks the GPU payload of the sampled
streams byte for byte against the CPU coder.  The always-on gate is the device-side round trip: decoded bins == input
bins for every stream, decoder integrity flags all zero.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s measured copy)
BASELINE_METRIC = "Gbin/s encode+decode round-trip, 1M streams×4K bins; bit-exact vs HM CPU"  # BASELINE.json, verbatim

WORKLOADS = {
    # name: global streams, per-GPU streams of the weak mode, bins per stream, contexts, SURVEY 8(d) seed, 50 % bypass
    "config3": dict(S=1 << 20, per_gpu=1 << 20, B=4096, C=64, seed=0xCABAC003, ep=False),
    "config5": dict(S=8 << 20, per_gpu=1 << 20, B=8192, C=64, seed=0xCABAC005, ep=True),
    "config2": dict(S=1 << 16, per_gpu=1 << 16, B=1024, C=3, seed=0xCABAC002, ep=False),
}
WORKLOADS["config5_per_gpu"] = dict(WORKLOADS["config5"], S=1 << 20)  # round-1 name: one GPU's share of config 5
CHUNK_STREAMS = 1 << 20  # streams per launch: 1 Mi x 8 Ki bins = 8 GiB per array, so a whole config-5 share fits 288 GB


def _cpu_worker(seed, B, C, ep, budget_s, s_first, s_step, tag, out):
    """One host thread: reference (or port) round trips on its own streams until the budget is spent."""
    import numpy as np
    from oracle import pyoracle
    orc = pyoracle.Oracle()
    init = np.zeros(C, np.uint8)
    use_ref = pyoracle.have_reference()
    ref = pyoracle.Reference(tag) if use_ref else None
    chunk = 256
    done, t_enc, t_dec, s0 = 0, 0.0, 0.0, s_first
    t_start = time.perf_counter()
    while time.perf_counter() - t_start < budget_s:
        bins, ctx = orc.synth(seed, s0, chunk, B, C, ep)
        s0 += s_step
        if use_ref:
            _tot, te, td = ref.roundtrip_batch(bins, ctx, init, decode=True)
        else:
            t0 = time.perf_counter()
            payload, offsets = orc.encode_batch(bins, ctx, init)
            t1 = time.perf_counter()
            dec, status = orc.decode_batch(payload, offsets, ctx, init)
            t2 = time.perf_counter()
            assert (dec == bins).all() and (status == 0).all()
            te, td = t1 - t0, t2 - t1
        t_enc += te
        t_dec += td
        done += chunk
    out.append((done, t_enc, t_dec, time.perf_counter() - t_start))


def cpu_baseline(seed, B, C, ep, budget_s, gpu_payload_of=None, nchk=4096):
    """Reference (or port) round trip on a bounded sample of the same synthetic workload: one host core (the headline
    `value`, like the single-threaded reference), then all host cores -- the reference classes with one scratch file per
    stream (what the reference does), and the in-memory C restatement on native threads (SURVEY 8(d)'s CPU ceiling).
    gpu_payload_of(s) -> bytes of GPU-coded stream s: the first 4 096 streams (SURVEY 8(d)'s sampled subset) are also coded
    by the CPU restatement here and must match the GPU payload byte for byte (the oracle stays inside this leg)."""
    import threading
    import numpy as np
    from oracle import pyoracle
    use_ref = pyoracle.have_reference()
    checked = 0
    orc = pyoracle.Oracle()
    if gpu_payload_of is not None:
        hb, hc = orc.synth(seed, 0, nchk, B, C, ep)
        init = np.zeros(C, np.uint8)
        for s in range(nchk):
            want = orc.encode_stream(hb[s], hc[s], init).tobytes()
            assert gpu_payload_of(s) == want, "GPU payload of stream %d differs from the CPU coder" % s
            checked += 1
    res = []
    _cpu_worker(seed, B, C, ep, budget_s * 0.4, 0, 256, "ref1", res)
    done, t_enc, t_dec, _wall = res[0]
    total_bins = done * B
    out = {
        "value": total_bins / (t_enc + t_dec) / 1e9,
        "unit": "Gbin/s",
        "cores": 1,
        "kind": "reference" if use_ref else "port",
        "sample": "%d streams x %d bins of the same generator (seed 0x%X), encode+decode, "
                  "%s" % (done, B, seed, "reference classes, one scratch file per stream on /dev/shm"
                          if use_ref else "C restatement, in-memory sink"),
        "encode_Gbin_s": total_bins / t_enc / 1e9,
        "decode_Gbin_s": total_bins / t_dec / 1e9,
        "gpu_streams_checked_byte_exact": checked,
    }
    ncores = os.cpu_count() or 1
    if ncores > 1:
        res, threads = [], []
        for i in range(ncores):  # ctypes calls release the GIL: real parallelism
            th = threading.Thread(target=_cpu_worker, args=(seed, B, C, ep, budget_s * 0.3, 1 << 20 | i * 256, ncores * 256,
                                                            "refm%d" % i, res))
            th.start()
            threads.append(th)
        for th in threads:
            th.join()
        rate = sum(r[0] * B / (r[1] + r[2]) for r in res)  # coding time only, like the single-core figure
        out["all_cores"] = {"value": rate / 1e9, "unit": "Gbin/s", "cores": ncores,
                            "note": "sum of the coding rates of %d concurrent threads, each with its own streams and "
                                    "scratch file" % ncores}
        if hasattr(orc, "roundtrip_threads"):
            # the restatement with an in-memory sink on native threads, streams split evenly: no files, no Python between
            # streams -- the honest all-core CPU figure
            nstreams, wall, ok = orc.roundtrip_threads(seed, 1 << 21, B, C, ep, ncores, budget_s * 0.3)
            assert ok, "in-memory restatement round trip failed"
            out["all_cores_in_memory"] = {"value": nstreams * B / wall / 1e9, "unit": "Gbin/s", "cores": ncores,
                                          "kind": "port",
                                          "note": "C restatement, in-memory sink, %d native threads for %.1f s, %d streams "
                                                  "completed" % (ncores, wall, nstreams)}
    return out


def launch_ranks(n):
    """`python bench.py --gpus N` with no launcher around it: start the N ranks as CHILD processes of this one -- which
    has not touched the GPU and never replaces itself -- under the same launcher the driver uses (torch.distributed.run,
    one rank per GPU, rendezvous on 127.0.0.1), pass their output through (rank 0 prints the one JSON line) and return
    the launcher's exit status."""
    import socket
    import subprocess
    port = os.environ.get("MASTER_PORT")
    if not port:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = str(sk.getsockname()[1])
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    env.pop("ISSCABAC_SELF_LAUNCH", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr",
           "127.0.0.1", "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)  # 0.3 s of timed region: box noise averages out better than over 20
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="config3", choices=sorted(WORKLOADS))
    ap.add_argument("--scaling", default="strong", choices=("strong", "weak"))
    ap.add_argument("--streams", type=int, default=0, help="override the workload's stream count (global in strong "
                                                           "mode, per GPU in weak mode)")
    ap.add_argument("--chunk-streams", type=int, default=CHUNK_STREAMS, help="streams per kernel launch")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the CPU baseline leg (0 = skip)")
    ap.add_argument("--tuning", default="auto", choices=("auto", "throughput", "latency", "latency_single"),
                    help="kernel instance (isscabac_set_tuning); auto picks by launch size")
    ap.add_argument("--round-trip", action="store_true", help="config2: time encode + decode instead of its encode-only step")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--no-gather", action="store_true", help="skip the final bitstream gather (N > 1)")
    ap.add_argument("--rendezvous-only", action="store_true",
                    help="launch check without a GPU: the ranks meet over gloo, reduce one number and rank 0 prints it")
    args = ap.parse_args()

    if "RANK" not in os.environ and (args.gpus > 1 or os.environ.get("ISSCABAC_SELF_LAUNCH")):
        sys.exit(launch_ranks(args.gpus))  # nothing in this process has touched the GPU (torch is not even imported)

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, "--gpus %d but WORLD_SIZE=%d: the launcher's --nproc-per-node must equal --gpus" % (args.gpus, world)
    if args.rendezvous_only:
        dist.init_process_group("gloo")
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        dist.barrier()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        if rank == 0:
            print(json.dumps({"rendezvous_only": True, "world": world, "max_rank_plus_1": t.item()}))
        dist.barrier()
        dist.destroy_process_group()
        return
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # ISSCABAC_FORCE_DIST=1: take the multi-rank code path (RCCL init, barriers, reduction, gather) with one rank --
    # the only way to exercise it on a 1-GPU box
    multi = world > 1 or bool(os.environ.get("ISSCABAC_FORCE_DIST"))
    if multi:
        dist.init_process_group("nccl", device_id=dev)

    from isscabac_amd import Engine, sharding, stream_bound
    wl = dict(WORKLOADS[args.workload])
    if args.streams:
        wl["S"] = wl["per_gpu"] = args.streams
    B, C, seed, ep = wl["B"], wl["C"], wl["seed"], wl["ep"]
    encode_only = args.workload == "config2" and not args.round_trip  # BASELINE.json: "64K streams x 1K bins, 3 contexts, encode-only"
    chunks = sharding.plan(wl["S"], wl["per_gpu"], rank, world, args.scaling, args.chunk_streams)
    job_streams = sharding.job_streams(wl["S"], wl["per_gpu"], world, args.scaling)
    my_streams = sum(hi - lo for lo, hi in chunks)
    eng = Engine(local_rank)
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    eng.set_profiling(True)
    eng.set_tuning(args.tuning)

    # ---- synthetic inputs, resident in HBM before anything is timed ----------------------------------
    init = np.zeros(C, np.uint8)  # (mps 0, state 0) == initByProb(0.5), cabacDemo.m:49
    bufs = []
    for lo, hi in chunks:
        S = hi - lo
        cap = S * (B // 8 + B // 16 + 64)  # 1.5 bit/bin: EP-heavy streams stay below ~1.01
        if os.environ.get("ISSCABAC_BENCH_FULL_CAP"):  # timing-only ablation builds emit garbage of any size
            cap = S * stream_bound(B)
        b = dict(S=S, lo=lo, cap=cap,
                 bins=torch.empty((S, B), dtype=torch.uint8, device=dev),
                 ctx=torch.empty((S, B), dtype=torch.uint8, device=dev),
                 payload=torch.empty(cap, dtype=torch.uint8, device=dev),
                 offsets=torch.empty(S + 1, dtype=torch.int64, device=dev),
                 out=torch.empty((S, B), dtype=torch.uint8, device=dev),
                 status=torch.empty((S,), dtype=torch.int32, device=dev))
        eng.synth_dev(seed, lo, S, B, B, C, ep, b["bins"].data_ptr(), b["ctx"].data_ptr())
        bufs.append(b)

    def step():
        # encode (kernel + offsets scan + payload gather) and decode are only enqueued: the GPU runs the launches back to
        # back, the host reads nothing in between (the decoder bounds its reads by the buffer's capacity; streams are
        # delimited by the offsets on the device).  Sizes, capacity check and kernel times are collected after the fence.
        for b in bufs:
            eng.encode_dev_async(b["bins"].data_ptr(), b["ctx"].data_ptr(), b["S"], B, B, init, b["payload"].data_ptr(),
                                 b["cap"], b["offsets"].data_ptr())
            if not encode_only:
                decode(b)

    def decode(b):
        eng.decode_dev_async(b["payload"].data_ptr(), b["cap"], b["offsets"].data_ptr(), b["ctx"].data_ptr(), b["S"], B, B, init,
                             b["out"].data_ptr(), b["status"].data_ptr())

    def fence():
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    eng.sync()
    eng.kernel_ms_mean()  # forget the warm-up launches
    eng.kernel_ms_samples()
    eng.kernel_ms_samples_dropped()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    eng.sync()  # raises if a payload buffer was too small (reports the most recent encode; the others are checked below)
    ksamples = eng.kernel_ms_samples()  # every launch of the timed steps (the same events the means below come from)
    kdropped = eng.kernel_ms_samples_dropped()  # non-zero: more launches than the engine keeps -- the series is truncated
    kms, klaunches = eng.kernel_ms_mean()  # HIP events on the launch stream around every kernel of the timed steps
    assert klaunches["encode"] == args.steps * len(bufs) and klaunches["decode"] == (0 if encode_only else args.steps * len(bufs))
    if encode_only:  # the decode rate and the gate below still need the way back: one pass, outside the timed region
        for b in bufs:
            decode(b)
        fence()
        eng.sync()
        kms_d, _ = eng.kernel_ms_mean()
        kms["decode"] = kms_d["decode"]
    # kernel time of every step (all launches of the step, all four kernels): min / median beside the mean, so that two
    # builds can be compared from one run each (box-to-box spread is +- 3 %, within a run the steps agree to ~0.5 %)
    nl = len(bufs)
    step_ms = []
    step_kernels = ("encode", "scan", "gather") if encode_only else ("encode", "scan", "gather", "decode")
    if all(len(ksamples[k]) == args.steps * nl for k in step_kernels):
        for i in range(args.steps):
            step_ms.append(sum(sum(ksamples[k][i * nl:(i + 1) * nl]) for k in step_kernels))
    totals = [int(b["offsets"][-1]) for b in bufs]
    for b, tot in zip(bufs, totals):
        assert tot <= b["cap"], "payload buffer too small"
    # what each rank saw, for the line (a first N-GPU run must explain itself: which rank was slow, which kernels it ran)
    mine = {"rank": rank, "device": local_rank, "ms_per_step": elapsed / args.steps * 1e3, "streams": my_streams,
            "launches_per_step": len(bufs), "payload_bytes": int(sum(totals)),
            "encode_instance": eng.last_instance("encode"), "decode_instance": eng.last_instance("decode"),
            "kernels_ms": {k: round(v, 4) for k, v in kms.items()}}
    per_rank = [mine]
    dist_info = None
    if multi:
        per_rank = [None] * dist.get_world_size()
        dist.all_gather_object(per_rank, mine)
        dist_info = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "launcher_world_size": world}
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- correctness gate (outside the timed region): the round trip must reproduce every bin of every stream --------
    verified = None
    if not args.no_verify:
        for b in bufs:
            assert int(b["status"].abs().sum()) == 0, "decoder integrity flags set"
            assert torch.equal(b["out"], b["bins"]), "decoded bins differ from the input"
        verified = True

    # ---- final bitstream gather to rank 0 (north_star's only exchange), timed apart -----------------------
    gather_ms, gather_ok, gather_bytes = None, None, None
    if multi and not args.no_gather:
        # runs on a helper thread under a deadline: a stuck exchange must not cost the run its result line -- it costs
        # the run its exit status ("gather_ok": false, non-zero exit after the line is out)
        import threading
        box = {}

        def do_gather():
            try:
                torch.cuda.set_device(local_rank)  # the current device is per thread
                p_mine, o_mine = sharding.concat_launches([(b["payload"][:t], b["offsets"]) for b, t in zip(bufs, totals)])
                # untimed first pass over each rank's first stream: the first all-gather and the first send / receive between two
                # ranks set up RCCL's channels and load torch's copy kernels (tens of ms, once per process) -- not the exchange
                sharding.gather_payload(p_mine[:int(o_mine[1])], o_mine[:2], dist)
                fence()
                g0 = time.perf_counter()
                p_all, o_all = sharding.gather_payload(p_mine, o_mine, dist)
                fence()
                box["ms"] = (time.perf_counter() - g0) * 1e3
                if rank == 0:
                    assert int(o_all[-1]) == p_all.numel() and o_all.numel() == job_streams + 1
                    box["bytes"] = int(p_all.numel())
                del p_all, o_all
                box["ok"] = True
            except Exception as exc:
                sys.stderr.write("bitstream gather failed (rank %d): %r\n" % (rank, exc))

        th = threading.Thread(target=do_gather, daemon=True)
        th.start()
        th.join(float(os.environ.get("ISSCABAC_GATHER_DEADLINE_S", "120")))
        if th.is_alive():
            sys.stderr.write("bitstream gather did not finish before its deadline (rank %d)\n" % rank)
        gather_ok = bool(box.get("ok"))
        gather_ms, gather_bytes = box.get("ms"), box.get("bytes")

    rc = 0 if gather_ok in (None, True) else 3
    if rank == 0:
        bins_total = float(job_streams) * B * args.steps
        value = bins_total / elapsed / 1e9
        S_l = my_streams / float(len(bufs))  # streams and payload bytes of one launch (the chunks are equal-sized)
        P_l = sum(totals) / float(len(bufs))
        enc_ms, dec_ms = kms["encode"], kms["decode"]
        enc_bytes = 2.0 * S_l * B + P_l + 8.0 * S_l
        dec_bytes = 2.0 * S_l * B + P_l + 8.0 * S_l
        if enc_ms >= dec_ms or encode_only:
            kname, kbytes, kms_avg = "k_encode", enc_bytes, enc_ms
        else:
            kname, kbytes, kms_avg = "k_decode", dec_bytes, dec_ms
        achieved = kbytes / (kms_avg * 1e-3) / 1e9
        traffic, traffic_source = None, None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath) and not args.streams and bufs[0]["S"] == CHUNK_STREAMS:
            try:  # per-launch HBM bytes of this kernel on this launch shape, from the committed rocprofv3 --pmc passes
                ent = json.load(open(tpath)).get(args.workload.replace("_per_gpu", ""), {}).get(kname)
                if ent:
                    traffic, traffic_source = ent["hbm_bytes_per_launch"], ent.get("source")
            except Exception:
                traffic = None
        shape = "%d streams x %d bins, %d contexts%s" % (job_streams, B, C, ", 50% bypass" if ep else "")
        if args.workload == "config3" and job_streams == 1 << 20:
            metric = BASELINE_METRIC
        else:
            metric = "Gbin/s %s, %s; bit-exact vs HM CPU" % ("encode-only" if encode_only else "encode+decode round-trip", shape)
        line = {
            "metric": metric,
            "value": value,
            "unit": "Gbin/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "config": {"workload": "%s: %s in all (%s scaling), %d streams on each of %d GPU%s in %d launch%s of <= %d "
                                   "streams, SURVEY 8(d) generator seed 0x%X"
                                   % (args.workload, shape, args.scaling, my_streams, world, "s" if world > 1 else "",
                                      len(bufs), "es" if len(bufs) > 1 else "", args.chunk_streams, seed),
                       "streams_total": job_streams, "streams_per_gpu": my_streams, "launches_per_step": len(bufs),
                       "bins_per_stream": B, "contexts": C, "payload_bytes_per_gpu": int(sum(totals)),
                       "bits_per_bin": 8.0 * sum(totals) / (float(my_streams) * B)},
            "verified_bit_exact": verified,
            # one launch's bins over its kernel time (HIP events): encode = kernel + offsets scan + payload gather
            "encode_Gbin_s": S_l * B / ((kms["encode"] + kms["scan"] + kms["gather"]) * 1e-3) / 1e9,
            "decode_Gbin_s": S_l * B / (kms["decode"] * 1e-3) / 1e9 if kms.get("decode") else None,
            "step": "encode (kernel + scan + gather)" if encode_only else "encode (kernel + scan + gather) + decode",
            "kernels_ms": kms,
            "kernels_ms_min": {k: (min(v) if v else None) for k, v in ksamples.items()},
            "kernels_ms_median": {k: (sorted(v)[len(v) // 2] if v else None) for k, v in ksamples.items()},
            "step_kernel_ms_min": min(step_ms) if step_ms else None,
            "step_kernel_ms_median": sorted(step_ms)[len(step_ms) // 2] if step_ms else None,
            "kernel_samples_truncated": {k: v for k, v in kdropped.items() if v} or None,  # min / median then cover the first 65536 launches only
            "roofline": {"bound": "hbm", "kernel": kname, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                         "algorithmic_bytes_per_launch": kbytes, "ms_per_launch": kms_avg},
            "gather_ms": gather_ms,
            "gather_ok": gather_ok,
            "gather_bytes": gather_bytes,
            "gather_GBps": gather_bytes / (gather_ms * 1e-3) / 1e9 if gather_ms and gather_bytes else None,
            "dist": dist_info,    # the process group as torch.distributed reports it (None: one rank, no group)
            "per_rank": per_rank,  # every rank's own wall time per step, shard, kernel instances and kernel times
        }
        if args.cpu_seconds > 0:
            b0 = bufs[0]  # rank 0's first launch starts at global stream 0 in both modes
            nchk = min(4096, b0["S"])
            off = b0["offsets"][:nchk + 1].cpu().numpy().astype(np.uint64)
            head = b0["payload"][:int(off[-1])].cpu().numpy()  # the sampled streams' bytes in one copy

            def gpu_payload_of(s):
                return head[int(off[s]):int(off[s + 1])].tobytes()

            line["cpu_baseline"] = cpu_baseline(seed, B, C, ep, args.cpu_seconds,
                                                None if args.no_verify else gpu_payload_of, nchk)
        print(json.dumps(line))
        sys.stdout.flush()
    if rc:
        sys.stderr.flush()
        os._exit(rc)  # the exchange is stuck inside the process group: leave, visibly failed, without waiting for it
    if multi:
        dist.barrier()  # the other ranks wait here while rank 0 runs its CPU leg
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
