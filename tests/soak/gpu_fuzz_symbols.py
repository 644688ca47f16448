#!/usr/bin/env python3
"""Randomised soak of the symbol-level kernels (binariser + context selector fused with the coder) against the oracle's
restatement of cabacDemo.m / +coder/cabacEncode.m / cabacDecode.m (test infrastructure; run by hand on a GPU box):
    python tests/soak/gpu_fuzz_symbols.py [seconds=120] [seed=1]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch  # noqa: F401
from isscabac_amd import Engine, SymbolParams
from oracle.pyoracle import Oracle

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
eng, orc = Engine(0), Oracle()
t0, rounds, flagged = time.time(), 0, 0
while time.time() - t0 < budget:
    S = int(rng.choice([1, 2, 40, 65, 130]))
    R = int(rng.choice([1, 7, 109, 400]))
    demo = rng.random() < 0.3
    if demo:
        nq = int(rng.choice([2, 4, 8]))
        params, nctx = SymbolParams.demo(nq), 3
    else:
        nq = int(rng.choice([2, 4, 8, 16]))
        method = "TU" if rng.random() < 0.3 else "EG"
        k = 0 if method == "TU" else int(rng.choice([0, 0, 1, 2]))
        types = int(rng.integers(0, 32))
        nlbp = int(rng.choice([1, 2, 3]))
        params, nctx = SymbolParams.iss(nq, method, k, types, nlbp), 7 * nlbp + 2
    init = rng.integers(0, 126, nctx).astype(np.uint8)
    skew = rng.random() * 3 + 0.3
    syms = np.minimum((rng.exponential(skew, (S, R))).astype(np.int64), nq - 1).astype(np.uint8)
    ns = rng.integers(0, R + 1, S).astype(np.uint32) if rng.random() < 0.5 else None
    payload, offsets, nbins = eng.encode_symbols(syms, params, init, ns)
    for s in range(S):
        n = R if ns is None else int(ns[s])
        if demo:
            want, wnb = orc.demo_encode(syms[s, :n], nq, init)
        else:
            want, wnb = orc.iss_encode_column(syms[s, :n], nq, method, k, types, nlbp, init)
        assert payload[int(offsets[s]):int(offsets[s + 1])].tobytes() == want.tobytes(), ("payload", S, R, demo, s)
        assert nbins[s] == wnb
    dec, st = eng.decode_symbols(payload, offsets, R, params, init, ns)
    assert not st.any()
    for s in range(S):
        n = R if ns is None else int(ns[s])
        assert (dec[s, :n] == syms[s, :n]).all() and (dec[s, n:R] == 0).all()
    bad = payload.copy()
    if bad.size:
        kf = max(1, bad.size // 30)
        idx = rng.integers(0, bad.size, kf)
        bad[idx] ^= (1 << rng.integers(0, 8, kf)).astype(np.uint8)
        for s in range(0, S, 4):
            if offsets[s + 1] > offsets[s]:
                bad[int(offsets[s])] = 0xFF
        dec, st = eng.decode_symbols(bad, offsets, R, params, init, ns)
        for s in range(S):
            n = R if ns is None else int(ns[s])
            p = bad[int(offsets[s]):int(offsets[s + 1])]
            if demo:
                want, wst = orc.demo_decode(p, n, nq, init)
            else:
                want, wst = orc.iss_decode_column(p, n, nq, method, k, types, nlbp, init)
            assert st[s] == wst, ("corrupt status", S, R, demo, s, st[s], wst)
            assert (dec[s, :n] == (want & 0xFF)).all(), ("corrupt symbols", S, R, demo, s)  # engine symbols are uint8
        flagged += int((st != 0).sum())
    rounds += 1
print("symbol soak ok: %d rounds, %d corrupted streams flagged, %.0f s" % (rounds, flagged, time.time() - t0))
