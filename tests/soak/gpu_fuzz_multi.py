#!/usr/bin/env python3
"""Randomised soak of the one-process multi-engine path (isscabac_multi: 1..4 engines, all on the test box's one GPU)
against the oracle; test infrastructure, run by hand on a GPU box:   python tests/soak/gpu_fuzz_multi.py [seconds=120] [seed=1]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch  # noqa: F401
from isscabac_amd.engine import MultiEngine
from oracle.pyoracle import Oracle

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
orc = Oracle()
engines = {n: MultiEngine([0] * n) for n in (1, 2, 3, 4)}
t0, rounds, bins_total, flagged = time.time(), 0, 0, 0
while time.time() - t0 < budget:
    eng = engines[int(rng.integers(1, 5))]
    S = int(rng.choice([1, 2, 3, 5, 64, 65, 200, 777, 3001]))
    B = int(rng.choice([1, 16, 17, 100, 640, 1000, 4096]))
    Cn = int(rng.choice([1, 3, 23, 64, 100, 190, 255]))
    ep = float(rng.choice([0.0, 0.0, 0.1, 0.5]))
    init = rng.integers(0, 126, Cn).astype(np.uint8)
    ctx = rng.integers(0, Cn, (S, B)).astype(np.uint8)
    ctx[rng.random((S, B)) < ep] = 0xFF
    skew = rng.random(Cn + 1) ** int(rng.integers(1, 5))
    bins = (rng.random((S, B)) < skew[np.where(ctx == 0xFF, Cn, ctx)]).astype(np.uint8)
    nb = rng.integers(0, B + 1, S).astype(np.uint32) if rng.random() < 0.4 else None
    want_p, want_o = orc.encode_batch(bins, ctx, init, nb)
    got_p, got_o = eng.encode(bins, ctx, init, nb)
    assert np.array_equal(got_o, want_o), ("offsets", S, B, Cn, ep)
    assert got_p.tobytes() == want_p.tobytes(), ("payload", S, B, Cn, ep)
    dec, st = eng.decode(got_p, got_o, ctx, init, nb)
    assert not st.any(), ("status", S, B, Cn, ep)
    for s in range(S):
        n = B if nb is None else int(nb[s])
        assert np.array_equal(dec[s, :n], bins[s, :n]), ("bins", S, B, Cn, ep, s)
    if got_p.size:
        bad = got_p.copy()
        k = max(1, bad.size // 50)
        bad[rng.integers(0, bad.size, k)] ^= (1 << rng.integers(0, 8, k)).astype(np.uint8)
        dec, st = eng.decode(bad, got_o, ctx, init, nb)
        wdec, wst = orc.decode_batch(bad, want_o, ctx, init, nb)
        assert np.array_equal(st, wst), ("corrupt status", S, B, Cn, ep)
        for s in range(S):
            n = B if nb is None else int(nb[s])
            assert np.array_equal(dec[s, :n], wdec[s, :n]), ("corrupt bins", S, B, Cn, ep, s)
        flagged += int((st != 0).sum())
    rounds += 1
    bins_total += S * B
print("multi soak ok: %d rounds, %.1f M bins, %d corrupted streams flagged, %.0f s" % (rounds, bins_total / 1e6, flagged, time.time() - t0))
