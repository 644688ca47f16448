#!/usr/bin/env python3
"""Randomised soak of the batch kernels against the oracle (test infrastructure; run by hand on a GPU box, not collected
by pytest):   python tests/soak/gpu_fuzz.py [seconds=120] [seed=1]
Every round draws a shape (streams, bins, contexts, bypass share, skew, ragged lengths, init states), encodes on the GPU
through the host-buffer C ABI (a quarter of the rounds with the bins as bits), compares every stream's bytes and the
offsets with the oracle, decodes, compares bins and status; then corrupts the payload and compares the decoder's bins and status flags with the oracle's again."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch  # noqa: F401
from isscabac_amd import Engine
from oracle.pyoracle import Oracle

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
eng, orc = Engine(0), Oracle()
t0, rounds, bins_total, flagged = time.time(), 0, 0, 0
while time.time() - t0 < budget:
    S = int(rng.choice([1, 3, 63, 64, 65, 200, 777, 2500, 40000, 70000]))
    B = int(rng.choice([1, 5, 16, 17, 100, 640, 1000, 4096, 5000]))
    if S >= 40000:  # AUTO's own choice between 32 Ki and 128 Ki streams (k_encode_pair + k_decode<1,true,false>): short rows, the oracle is one core
        B = int(rng.choice([1, 16, 17, 40, 100, 130]))
    Cn = int(rng.choice([1, 3, 4, 5, 23, 64, 65, 128, 129, 150, 190, 253, 255]))
    # (auto would always pick the wave pairs at the small sizes; at 40 000 / 70 000 streams it is what a strong-scaled job's rank runs)
    eng.set_tuning("auto" if S >= 40000 else str(rng.choice(["throughput", "latency", "latency_single"])))
    ep = float(rng.choice([0.0, 0.0, 0.05, 0.5, 1.0]))
    init = rng.integers(0, 126, Cn).astype(np.uint8)
    ctx = rng.integers(0, Cn, (S, B)).astype(np.uint8)
    ctx[rng.random((S, B)) < ep] = 0xFF
    skew = rng.random(Cn + 1) ** int(rng.integers(1, 6))
    if rng.random() < 0.2:
        skew[:] = rng.choice([0.0, 1.0, 0.5])  # all-MPS / all-LPS-ish / incompressible
    bins = (rng.random((S, B)) < skew[np.where(ctx == 0xFF, Cn, ctx)]).astype(np.uint8)
    nb = None
    if rng.random() < 0.4:
        nb = rng.integers(0, B + 1, S).astype(np.uint32)
    want_p, want_o = orc.encode_batch(bins, ctx, init, nb)
    packed = rng.random() < 0.25  # the bins-as-bits entry points of the host path
    pitch = max(16, (B + 15) // 16 * 16)
    if packed:
        padded = np.zeros((S, pitch), np.uint8)
        padded[:, :B] = bins
        got_p, got_o = eng.encode_packed(np.packbits(padded, axis=1, bitorder="little"), ctx, init, nb)
    else:
        got_p, got_o = eng.encode(bins, ctx, init, nb)
    assert np.array_equal(got_o, want_o), ("offsets", S, B, Cn, ep, packed)
    assert got_p.tobytes() == want_p.tobytes(), ("payload", S, B, Cn, ep, packed)
    if packed:
        bits, st = eng.decode_packed(got_p, got_o, ctx, init, nb)
        dec = np.unpackbits(bits, axis=1, bitorder="little")[:, :B]
    else:
        dec, st = eng.decode(got_p, got_o, ctx, init, nb)
    wdec, wst = orc.decode_batch(want_p, want_o, ctx, init, nb)
    assert np.array_equal(st, wst) and not st.any(), ("status", S, B, Cn, ep)
    for s in range(S):
        n = B if nb is None else int(nb[s])
        assert np.array_equal(dec[s, :n], bins[s, :n]), ("bins", S, B, Cn, ep, s)
    # corruption: flip bits, decode again, the oracle must agree on bins and flags
    if got_p.size:
        bad = got_p.copy()
        k = max(1, bad.size // 50)
        idx = rng.integers(0, bad.size, k)
        bad[idx] ^= (1 << rng.integers(0, 8, k)).astype(np.uint8)
        dec, st = eng.decode(bad, got_o, ctx, init, nb)
        wdec, wst = orc.decode_batch(bad, want_o, ctx, init, nb)
        assert np.array_equal(st, wst), ("corrupt status", S, B, Cn, ep)
        for s in range(S):
            n = B if nb is None else int(nb[s])
            assert np.array_equal(dec[s, :n], wdec[s, :n]), ("corrupt bins", S, B, Cn, ep, s)
        flagged += int((st != 0).sum())
    rounds += 1
    bins_total += S * B
print("soak ok: %d rounds, %.1f M bins, %d corrupted streams flagged, %.0f s" % (rounds, bins_total / 1e6, flagged, time.time() - t0))
