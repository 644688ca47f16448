#!/usr/bin/env python3
"""Randomised soak of the length-bucketed launches (device API with per-stream bin counts on the device and at least 4 096
streams: cabac_order.hip + the PERM kernel instances) against the oracle; test infrastructure, run by hand on a GPU box:
    python tests/soak/gpu_fuzz_bucketed.py [seconds=120] [seed=1]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from isscabac_amd import Engine
from oracle.pyoracle import Oracle

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
eng, orc = Engine(0), Oracle()
dev = torch.device("cuda:0")
t0, rounds, bins_total, flagged = time.time(), 0, 0, 0
while time.time() - t0 < budget:
    S = int(rng.choice([4096, 4097, 5000, 6143, 9000]))
    B = int(rng.choice([16, 40, 144, 256, 528, 1000]))
    Cn = int(rng.choice([1, 3, 23, 64, 65, 100, 128, 129, 190, 255]))
    ep = float(rng.choice([0.0, 0.0, 0.1, 0.5]))
    init = rng.integers(0, 126, Cn).astype(np.uint8)
    ctx = rng.integers(0, Cn, (S, B)).astype(np.uint8)
    ctx[rng.random((S, B)) < ep] = 0xFF
    skew = rng.random(Cn + 1) ** int(rng.integers(1, 5))
    bins = (rng.random((S, B)) < skew[np.where(ctx == 0xFF, Cn, ctx)]).astype(np.uint8)
    mode = int(rng.integers(0, 3))
    nb = (rng.integers(0, B + 1, S) if mode == 0 else rng.integers(B // 2, B + 1, S) if mode == 1
          else np.where(rng.random(S) < 0.1, 0, B)).astype(np.uint32)
    want_p, want_o = orc.encode_batch(bins, ctx, init, nb)
    pitch = (B + 15) // 16 * 16
    tb = torch.zeros((S, pitch), dtype=torch.uint8, device=dev)
    tc = torch.zeros((S, pitch), dtype=torch.uint8, device=dev)
    tb[:, :B] = torch.from_numpy(bins).to(dev)
    tc[:, :B] = torch.from_numpy(ctx).to(dev)
    tn = torch.from_numpy(nb.astype(np.int32)).to(dev)
    cap = int(want_p.size) + 64
    payload = torch.zeros(cap, dtype=torch.uint8, device=dev)
    offsets = torch.zeros(S + 1, dtype=torch.int64, device=dev)
    total = eng.encode_dev(tb.data_ptr(), tc.data_ptr(), S, B, pitch, init, payload.data_ptr(), cap, offsets.data_ptr(),
                           d_nbins=tn.data_ptr())
    assert total == want_p.size and offsets.cpu().numpy().astype(np.uint64).tolist() == want_o.tolist(), ("offsets", S, B, Cn, ep)
    assert payload[:total].cpu().numpy().tobytes() == want_p.tobytes(), ("payload", S, B, Cn, ep)
    out = torch.full((S, pitch), 7, dtype=torch.uint8, device=dev)
    status = torch.full((S,), 9, dtype=torch.int32, device=dev)
    eng.decode_dev(payload.data_ptr(), total, offsets.data_ptr(), tc.data_ptr(), S, B, pitch, init, out.data_ptr(),
                   status.data_ptr(), d_nbins=tn.data_ptr())
    got = out.cpu().numpy()[:, :B]
    mask = np.arange(B)[None, :] < nb[:, None]
    assert int(status.abs().sum()) == 0 and (got * mask == bins * mask).all() and not (got * ~mask).any(), ("bins", S, B, Cn, ep)
    if want_p.size:
        bad = want_p.copy()
        bad[rng.integers(0, bad.size, max(1, bad.size // 40))] ^= 0x5A
        wd, ws = orc.decode_batch(bad, want_o, ctx, init, nb)
        payload[:total] = torch.from_numpy(bad).to(dev)
        eng.decode_dev(payload.data_ptr(), total, offsets.data_ptr(), tc.data_ptr(), S, B, pitch, init, out.data_ptr(),
                       status.data_ptr(), d_nbins=tn.data_ptr())
        st = status.cpu().numpy().astype(np.uint32)
        assert (st == ws).all() and (out.cpu().numpy()[:, :B] * mask == wd * mask).all(), ("corrupt", S, B, Cn, ep)
        flagged += int((st != 0).sum())
    rounds += 1
    bins_total += int(nb.sum())
print("bucketed soak ok: %d rounds, %.1f M bins, %d corrupted streams flagged, %.0f s" % (rounds, bins_total / 1e6, flagged, time.time() - t0))
