#!/usr/bin/env python3
"""Randomised soak of the 16-bit-context batch calls (isscabac_*_batch16: up to 999 contexts) against the compiled reference
(oracle/_ref, whose coder holds 999 context models) or, where that is absent, the restatement -- test infrastructure, run by
hand on a GPU box:   python tests/soak/gpu_fuzz_wide.py [seconds=120] [seed=1]
Every round draws a shape (streams, bins, contexts, bypass share, ragged lengths, init states), encodes on the GPU, compares
every stream's bytes with the oracle's op-script coder, decodes, compares bins and status; then corrupts the payload (a byte
flipped, or a leading 0xFF) and compares the decoder's bins and its status flags with the oracle's again."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch  # noqa: F401
from isscabac_amd import Engine
from oracle import pyoracle

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
eng, orc = Engine(0), pyoracle.Oracle()
ref = pyoracle.Reference("wsoak") if pyoracle.have_reference() else None
t0, rounds, bins_total, flagged = time.time(), 0, 0, 0
while time.time() - t0 < budget:
    S = int(rng.choice([1, 3, 64, 65, 130]))
    B = int(rng.choice([1, 16, 17, 100, 333]))
    Cn = int(rng.choice([1, 64, 255, 256, 257, 500, 999]))
    ep = float(rng.choice([0.0, 0.05, 0.5]))
    init = rng.integers(0, 126, Cn).astype(np.uint8)
    ctx = rng.integers(0, Cn, (S, B)).astype(np.uint16)
    ctx[rng.random((S, B)) < ep] = 0xFFFF
    skew = rng.random(Cn + 1) ** int(rng.integers(1, 5))
    bins = (rng.random((S, B)) < skew[np.where(ctx == 0xFFFF, Cn, ctx)]).astype(np.uint8)
    nb = rng.integers(0, B + 1, S).astype(np.uint32) if rng.random() < 0.4 else np.full(S, B, np.uint32)
    payload, offsets = eng.encode16(bins, ctx, init, nb)
    bad = payload.copy()
    scripts = []
    for s in range(S):
        n = int(nb[s])
        kind = (ctx[s, :n] == 0xFFFF).astype(np.uint8)
        arg = np.where(kind == 1, 0, ctx[s, :n]).astype(np.uint32)
        coder = ref if ref is not None else orc
        want = coder.encode_ops(kind, bins[s, :n].astype(np.uint32), arg, init)[0]
        lo, hi = int(offsets[s]), int(offsets[s + 1])
        assert payload[lo:hi].tobytes() == want.tobytes(), (rounds, s, S, B, Cn)
        scripts.append((kind, arg))
        if hi > lo:
            if rng.random() < 0.3:
                bad[lo] = 0xFF
            elif rng.random() < 0.5:
                bad[int(rng.integers(lo, hi))] ^= int(rng.integers(1, 256))
    dec, status = eng.decode16(payload, offsets, ctx, init, nb)
    assert (status == 0).all()
    for s in range(S):
        assert (dec[s, :nb[s]] == bins[s, :nb[s]]).all() and (dec[s, nb[s]:] == 0).all()
    dec, status = eng.decode16(bad, offsets, ctx, init, nb)
    for s in range(S):
        kind, arg = scripts[s]
        res = orc.decode_ops(bad[int(offsets[s]):int(offsets[s + 1])], kind, arg, init)  # the restatement: all three status flags
        assert (dec[s, :nb[s]] == res[0]).all() and int(status[s]) == int(res[1]), (rounds, s)
        flagged += int(status[s] != 0)
    rounds += 1
    bins_total += int(nb.sum())
print("wide soak ok: %d rounds, %.2f M bins, %d corrupted streams flagged, %.0f s" % (rounds, bins_total / 1e6, flagged, time.time() - t0))
