#!/usr/bin/env python3
"""Randomised soak of the per-bin handle (isscabac_stream_*: what the reference's 'encodeBin' / 'decodeBin' / 'getNumBits'
commands map to) against the oracle; test infrastructure, run by hand on a GPU box:
    python tests/soak/gpu_fuzz_handle.py [seconds=120] [seed=1]
Every round draws one stream (context bins and bypass bins, random initial states, random skew), feeds it to a handle in
calls of random size (1 bin .. the whole rest), compares the bytes with the oracle's, decodes it the same way, then decodes a
corrupted copy and compares bins and status flags with the oracle's."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch  # noqa: F401
from isscabac_amd import Engine
from isscabac_amd.engine import Stream
from oracle.pyoracle import Oracle

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
eng, orc = Engine(0), Oracle()


def pieces(n):
    i = 0
    while i < n:
        k = int(rng.choice([1, 1, 2, 3, 7, 16, 100, 1000, n]))
        yield i, min(n, i + k)
        i += k


t0, rounds, bins_total, flagged = time.time(), 0, 0, 0
while time.time() - t0 < budget:
    n = int(rng.choice([0, 1, 2, 15, 16, 17, 200, 1000, 5000]))
    Cn = int(rng.choice([1, 3, 23, 64, 128, 200, 255]))
    ep = float(rng.choice([0.0, 0.1, 0.5]))
    init = rng.integers(0, 126, Cn).astype(np.uint8)
    ctx = rng.integers(0, Cn, n).astype(np.uint8)
    ctx[rng.random(n) < ep] = 0xFF
    skew = rng.random(Cn + 1) ** int(rng.integers(1, 5))
    bins = (rng.random(n) < skew[np.where(ctx == 0xFF, Cn, ctx)]).astype(np.uint8)
    want = orc.encode_stream(bins, ctx, init)
    h = Stream(eng, init)
    h.encode_start()
    for a, b in pieces(n):
        h.encode_bins(bins[a:b], ctx[a:b])
        if rng.random() < 0.05:
            h.num_bits()
    got = h.encode_finish()
    assert got.tobytes() == want.tobytes(), ("payload", n, Cn, ep)
    h.decode_start(got)
    dec = np.concatenate([h.decode_bins(ctx[a:b]) for a, b in pieces(n)]) if n else np.zeros(0, np.uint8)
    st = h.decode_finish()
    assert st == 0 and np.array_equal(dec, bins), ("decode", n, Cn, ep, st)
    if want.size and n:
        bad = want.copy()
        bad[int(rng.integers(0, bad.size))] ^= 1 << int(rng.integers(0, 8))
        wdec, wst = orc.decode_stream(bad, ctx, init)
        # (a fresh handle: like the reference's, a handle's context models are set at creation only -- a second decodeStart
        # on the same handle would go on with the states the first decode left, ref: SimpleCABACMex.cpp:165-176, 280-300)
        h.close()
        h = Stream(eng, init)
        h.decode_start(bad)
        dec = np.concatenate([h.decode_bins(ctx[a:b]) for a, b in pieces(n)])
        st = h.decode_finish()
        assert st == wst and np.array_equal(dec, wdec), ("corrupt", n, Cn, ep, st, wst)
        flagged += int(st != 0)
    h.close()
    rounds += 1
    bins_total += n
print("handle soak ok: %d rounds, %.2f M bins, %d corrupted streams flagged, %.0f s" % (rounds, bins_total / 1e6, flagged, time.time() - t0))
