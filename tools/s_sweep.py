#!/usr/bin/env python3
"""Single-GPU stream-count sweep = the predicted strong-scaling curve of config 3.

Streams never communicate, so N GPUs splitting 1 Mi streams each run the engine at S = 1 Mi / N streams: the strong-scaling
efficiency at N GPUs is rate(1 Mi / N) / rate(1 Mi) measured on ONE GPU (the final bitstream gather is timed apart, as in
bench.py).  Same step as bench.py (asynchronous encode + decode, inputs resident, wall time between two fences).

    python tools/s_sweep.py [--bins 4096] [--ctx 64] [--steps 20] [--out gpurun_out/s_sweep.json]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from isscabac_amd import Engine


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bins", type=int, default=4096)
    ap.add_argument("--ctx", type=int, default=64)
    ap.add_argument("--seed", type=lambda x: int(x, 0), default=0xCABAC003)
    ap.add_argument("--ep", action="store_true")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--streams", default="1048576,524288,262144,131072,65536,32768")
    ap.add_argument("--tuning", default="auto", choices=("auto", "throughput", "latency", "latency_single"))
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    eng = Engine(0)
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    eng.set_tuning(args.tuning)
    B, C = args.bins, args.ctx
    init = np.zeros(C, np.uint8)
    rows = []
    for S in [int(x) for x in args.streams.split(",")]:
        bins = torch.empty((S, B), dtype=torch.uint8, device=dev)
        ctx = torch.empty((S, B), dtype=torch.uint8, device=dev)
        eng.synth_dev(args.seed, 0, S, B, B, C, args.ep, bins.data_ptr(), ctx.data_ptr())
        cap = S * (B // 8 + B // 16 + 64)
        payload = torch.empty(cap, dtype=torch.uint8, device=dev)
        offsets = torch.empty(S + 1, dtype=torch.int64, device=dev)
        out = torch.empty((S, B), dtype=torch.uint8, device=dev)
        status = torch.empty((S,), dtype=torch.int32, device=dev)

        def step():
            eng.encode_dev_async(bins.data_ptr(), ctx.data_ptr(), S, B, B, init, payload.data_ptr(), cap, offsets.data_ptr())
            eng.decode_dev_async(payload.data_ptr(), cap, offsets.data_ptr(), ctx.data_ptr(), S, B, B, init, out.data_ptr(),
                                 status.data_ptr())

        # small launches: more steps so that the timed region stays well above the fence cost
        steps = max(args.steps, min(400, int(args.steps * (1 << 20) / S)))
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        eng.sync()
        eng.set_profiling(False)  # the event ring holds 128 launches: keep it out of a region of up to 800
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        eng.sync()
        eng.set_profiling(True)  # kernel times from a short run of their own
        eng.kernel_ms_mean()
        for _ in range(20):
            step()
        eng.sync()
        kms, _ = eng.kernel_ms_mean()
        assert int(status.abs().sum()) == 0 and torch.equal(out, bins)
        rate = S * B * steps / el / 1e9
        krate = S * B / ((kms["encode"] + kms["scan"] + kms["gather"] + kms["decode"]) * 1e-3) / 1e9
        rows.append(dict(tuning=args.tuning, streams=S, bins=B, contexts=C, steps=steps, ms_per_step=el / steps * 1e3, Gbin_s=rate,
                         kernels_only_Gbin_s=krate, kernels_ms=kms, waves_per_simd=S / 64.0 / 1024.0))
        del bins, ctx, payload, offsets, out, status
    base = rows[0]["Gbin_s"]
    for r in rows:
        r["n_gpus_equiv"] = rows[0]["streams"] // r["streams"]
        r["predicted_strong_scaling_efficiency"] = r["Gbin_s"] / base
        print("%8d streams x %d bins (%5.2f waves/SIMD): %7.3f ms/step  %6.1f Gbin/s (kernels only %6.1f)  enc %.3f dec %.3f ms"
              "  -> efficiency at %d GPUs %.3f" % (r["streams"], B, r["waves_per_simd"], r["ms_per_step"], r["Gbin_s"],
                                                   r["kernels_only_Gbin_s"], r["kernels_ms"]["encode"],
                                                   r["kernels_ms"]["decode"], r["n_gpus_equiv"],
                                                   r["predicted_strong_scaling_efficiency"]))
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        json.dump(rows, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
