#!/usr/bin/env python3
"""bench.py -- encode+decode round-trip throughput of the batched CABAC engine on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch of synthetic bin streams: encode (kernel + offsets
scan + payload gather) followed by decode, inputs already resident in HBM.  The steps are enqueued through the
engine's asynchronous entry points (isscabac_*_batch_dev_async): the host reads nothing between kernels, so the
figure does not depend on host wake-up latency; sizes, the capacity check and the kernel times are read after the
closing fence.  Workload = BASELINE.json
config 3 (1 Mi streams x 4 Ki bins, 64 contexts, SURVEY.md 8(d) generator) per GPU; with N > 1 every rank
codes its own shard of the global stream range (streams are independent: no data-path collective,
"weak" scaling), and the final bitstream gather to rank 0 is timed separately (gather_ms), outside `value`.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (the longer of the encode / decode
kernels), its duration measured with HIP events on the launch stream inside the engine; algorithmic bytes
follow SURVEY.md 8(d): encode reads 2 B/bin and writes P + 8 B/stream, decode reads 1 B/bin + P + 8 B/stream
and writes 1 B/bin (P = measured payload bytes).  `cpu_baseline` times the real reference classes
(oracle/_ref, kind "reference") or, where that build is absent, the C restatement (kind "port") on a bounded
sample of the same workload on the host cores of this box, and -- the only place bench.py touches oracle/ -- checks
the GPU payload of the sampled streams byte for byte against the CPU coder.  The always-on gate is the device-side
round trip: decoded bins == input bins for every stream, decoder integrity flags all zero.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s measured copy)

WORKLOADS = {
    # name: (S, B, C, seed, ep_half)
    "config3": (1 << 20, 4096, 64, 0xCABAC003, False),
    "config2": (1 << 16, 1024, 3, 0xCABAC002, False),
    "config5_per_gpu": (1 << 20, 8192, 64, 0xCABAC005, True),
}


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
    `value`, like the single-threaded reference), then all host cores with the streams split evenly.
    gpu_payload_of(s) -> bytes of GPU-coded stream s: the first 4 096 streams (SURVEY 8(d)'s sampled subset) are also coded
    by the CPU restatement here and must match the GPU payload byte for byte (the oracle stays inside this leg)."""
    import threading
    import numpy as np
    from oracle import pyoracle
    use_ref = pyoracle.have_reference()
    checked = 0
    if gpu_payload_of is not None:
        orc = pyoracle.Oracle()
        hb, hc = orc.synth(seed, 0, nchk, B, C, ep)
        init = np.zeros(C, np.uint8)
        for s in range(nchk):
            want = orc.encode_stream(hb[s], hc[s], init).tobytes()
            assert gpu_payload_of(s) == want, "GPU payload of stream %d differs from the CPU coder" % s
            checked += 1
    res = []
    _cpu_worker(seed, B, C, ep, budget_s * 0.5, 0, 256, "ref1", res)
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
            th = threading.Thread(target=_cpu_worker, args=(seed, B, C, ep, budget_s * 0.5, 1 << 20 | i * 256, ncores * 256,
                                                            "refm%d" % i, res))
            th.start()
            threads.append(th)
        for th in threads:
            th.join()
        rate = sum(r[0] * B / (r[1] + r[2]) for r in res)  # coding time only, like the single-core figure
        out["all_cores"] = {"value": rate / 1e9, "unit": "Gbin/s", "cores": ncores,
                            "note": "sum of the coding rates of %d concurrent threads, each with its own streams and "
                                    "scratch file" % ncores}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="config3", choices=sorted(WORKLOADS))
    ap.add_argument("--streams", type=int, default=0, help="override streams per GPU (debug)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the CPU baseline leg (0 = skip)")
    ap.add_argument("--no-verify", action="store_true")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # ISSCABAC_FORCE_DIST=1: take the multi-rank code path (RCCL init, barriers, reduction, gather) with one rank --
    # the only way to exercise it on a 1-GPU box
    multi = world > 1 or bool(os.environ.get("ISSCABAC_FORCE_DIST"))
    if multi:
        dist.init_process_group("nccl", device_id=dev)

    from isscabac_amd import Engine, stream_bound
    S, B, C, seed, ep = WORKLOADS[args.workload]
    if args.streams:
        S = args.streams
    eng = Engine(local_rank)
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    eng.set_profiling(True)

    # ---- synthetic inputs, resident in HBM before anything is timed ----------------------------------
    s0 = rank * S  # this rank's shard of the global stream range
    bins = torch.empty((S, B), dtype=torch.uint8, device=dev)
    ctx = torch.empty((S, B), dtype=torch.uint8, device=dev)
    eng.synth_dev(seed, s0, S, B, B, C, ep, bins.data_ptr(), ctx.data_ptr())
    init = np.zeros(C, np.uint8)  # (mps 0, state 0) == initByProb(0.5), cabacDemo.m:49
    cap = S * (B // 8 + B // 16 + 64)  # 1.5 bit/bin: EP-heavy streams stay below ~1.01
    if os.environ.get("ISSCABAC_BENCH_FULL_CAP"):  # timing-only ablation builds emit garbage of any size
        cap = S * stream_bound(B)
    payload = torch.empty(cap, dtype=torch.uint8, device=dev)
    offsets = torch.empty(S + 1, dtype=torch.int64, device=dev)
    out = torch.empty((S, B), dtype=torch.uint8, device=dev)
    status = torch.empty((S,), dtype=torch.int32, device=dev)

    def step():
        # encode (kernel + offsets scan + payload gather) and decode are only enqueued: the GPU runs the steps back to back,
        # the host reads nothing in between (the decoder bounds its reads by the buffer's capacity; streams are delimited by
        # the offsets on the device).  Payload size, capacity check and kernel times are collected after the fence.
        eng.encode_dev_async(bins.data_ptr(), ctx.data_ptr(), S, B, B, init, payload.data_ptr(), cap, offsets.data_ptr())
        eng.decode_dev_async(payload.data_ptr(), cap, offsets.data_ptr(), ctx.data_ptr(), S, B, B, init, out.data_ptr(),
                             status.data_ptr())

    def fence():
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    eng.sync()
    eng.kernel_ms_mean()  # forget the warm-up launches
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    total = eng.sync()  # payload bytes of the last step; raises if the payload buffer was too small
    kms, klaunches = eng.kernel_ms_mean()  # HIP events on the launch stream around every kernel of the timed steps
    assert klaunches["encode"] == args.steps and klaunches["decode"] == args.steps
    if multi:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- correctness gate (outside the timed region): the round trip must reproduce every bin of every stream --------
    verified = None
    if not args.no_verify:
        assert int(status.abs().sum()) == 0, "decoder integrity flags set"
        assert torch.equal(out, bins), "decoded bins differ from the input"
        verified = True

    # ---- final bitstream gather to rank 0 (north_star's only exchange), timed apart -----------------------
    gather_ms = None
    gather_hung = False
    if multi:
        # runs on a helper thread under a deadline: a stuck exchange must cost the run its gather_ms, not its result line
        import threading
        box = {}

        def do_gather():
            try:
                from isscabac_amd import sharding
                torch.cuda.set_device(local_rank)  # the current device is per thread
                fence()
                g0 = time.perf_counter()
                p_all, o_all = sharding.gather_payload(payload[:total], offsets, dist)
                fence()
                box["ms"] = (time.perf_counter() - g0) * 1e3
                if rank == 0:
                    assert int(o_all[-1]) == p_all.numel() and o_all.numel() == world * S + 1
                del p_all, o_all
            except Exception as exc:  # the gather is reported apart from `value`; never let it take the run down
                sys.stderr.write("bitstream gather failed: %r\n" % (exc,))

        th = threading.Thread(target=do_gather, daemon=True)
        th.start()
        th.join(float(os.environ.get("ISSCABAC_GATHER_DEADLINE_S", "60")))
        gather_hung = th.is_alive()
        if gather_hung:
            sys.stderr.write("bitstream gather did not finish before its deadline (rank %d)\n" % rank)
        gather_ms = box.get("ms")

    if rank == 0:
        bins_total = float(S) * B * world * args.steps
        value = bins_total / elapsed / 1e9
        P = float(total)
        enc_ms, dec_ms = kms["encode"], kms["decode"]
        enc_bytes = 2.0 * S * B + P + 8.0 * S
        dec_bytes = 2.0 * S * B + P + 8.0 * S
        if enc_ms >= dec_ms:
            kname, kbytes, kms_avg = "k_encode", enc_bytes, enc_ms
        else:
            kname, kbytes, kms_avg = "k_decode", dec_bytes, dec_ms
        achieved = kbytes / (kms_avg * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                ent = tj.get(args.workload, {}).get(kname)
                if ent and not args.streams:
                    traffic = ent["hbm_bytes_per_launch"]
            except Exception:
                traffic = None
        line = {
            "metric": "Gbin/s encode+decode round-trip, 1M streams\u00d74K bins; bit-exact vs HM CPU",  # BASELINE.json, verbatim
            "value": value,
            "unit": "Gbin/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "config": {"workload": "%s: %d streams x %d bins per GPU, %d contexts%s, SURVEY 8(d) generator seed 0x%X"
                                   % (args.workload, S, B, C, ", 50% bypass" if ep else "", seed),
                       "streams_per_gpu": S, "bins_per_stream": B, "contexts": C,
                       "payload_bytes_per_gpu": int(total), "bits_per_bin": 8.0 * P / (float(S) * B)},
            "verified_bit_exact": verified,
            "kernels_ms": kms,
            "roofline": {"bound": "hbm", "kernel": kname, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": kbytes, "ms_per_launch": kms_avg},
            "gather_ms": gather_ms,
        }
        if args.cpu_seconds > 0 and world == 1:
            nchk = min(4096, S)
            off = offsets[:nchk + 1].cpu().numpy().astype(np.uint64)
            head = payload[:int(off[-1])].cpu().numpy()  # the sampled streams' bytes in one copy

            def gpu_payload_of(s):
                return head[int(off[s]):int(off[s + 1])].tobytes()

            line["cpu_baseline"] = cpu_baseline(seed, B, C, ep, args.cpu_seconds,
                                                None if args.no_verify else gpu_payload_of, nchk)
        elif args.cpu_seconds > 0:
            line["cpu_baseline"] = None
        print(json.dumps(line))
    if gather_hung:
        sys.stdout.flush()
        os._exit(0)  # the exchange is stuck inside the process group: leave without waiting for it
    if multi:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
