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
    # --- whole matrix as ONE stream (col_rows), empirical context init, per-bin handle: a smaller case each round
    if not demo and rounds % 4 == 0:
        rows, cols = int(rng.choice([1, 5, 20, 109])), int(rng.choice([1, 3, 20]))
        G = np.minimum(rng.exponential(skew, (rows, cols)).astype(np.int64), nq - 1).astype(np.uint8)
        mp = SymbolParams.iss(nq, method, k, types, nlbp, col_rows=rows)
        pm, om, nbm = eng.encode_symbols(G.T.reshape(1, -1).copy(), mp, init)
        want, wnb = orc.iss_encode_matrix(G, nq, method, k, types, nlbp, init)
        assert pm.tobytes() == want.tobytes() and nbm[0] == wnb, ("matrix payload", rows, cols)
        dm, sm = eng.decode_symbols(pm, om, rows * cols, mp, init)
        assert sm[0] == 0 and (dm[0, :rows * cols].reshape(cols, rows).T == G).all()
        bm = pm.copy()
        bm[int(rng.integers(0, bm.size))] ^= 1 << int(rng.integers(0, 8))
        if rounds % 8 == 0:
            bm[0] = 0xFF
        dm, sm = eng.decode_symbols(bm, om, rows * cols, mp, init)
        wantm, wsm = orc.iss_decode_matrix(bm, rows, cols, nq, method, k, types, nlbp, init)
        assert sm[0] == wsm and (dm[0, :rows * cols].reshape(cols, rows).T == (wantm & 0xFF)).all(), ("matrix corrupt", rows, cols)
        _, _, _, hist, cost, hb = eng.encode_symbols_stats(G.T.reshape(1, -1).copy(), mp, init)
        _, wh, wc, wH = orc.iss_encode_matrix_stats(G, nq, method, k, types, nlbp, init)
        assert (hist[0, :nctx] == wh).all() and (cost[0, :nctx] == wc).all(), ("stats", rows, cols)
        assert (hb[0, :rows * cols].reshape(cols, rows).T == wH).all(), ("heat map", rows, cols)
        p0, b0 = eng.iss_init_context_model(G.T.copy(), SymbolParams.iss(nq, method, k, types, nlbp))
        wp0, wb0 = orc.iss_init_context_model(G, nq, method, k, types, nlbp)
        assert np.allclose(p0, wp0, rtol=0, atol=1e-15, equal_nan=True) and (b0 == wb0).all(), ("init model", rows, cols)
    if rounds % 16 == 0 and payload.size:
        from isscabac_amd import Stream
        n = int(rng.integers(1, 200))
        cx = rng.integers(0, nctx, n).astype(np.uint8)
        cx[rng.random(n) < 0.2] = 0xFF
        bn = rng.integers(0, 2, n).astype(np.uint8)
        h = Stream(eng, init)
        h.encode_start()
        i = 0
        while i < n:
            j = min(n, i + int(rng.integers(1, 9)))
            h.encode_bins(bn[i:j], cx[i:j])
            i = j
        ph = h.encode_finish()
        assert ph.tobytes() == orc.encode_stream(bn, cx, init).tobytes()
        bh = ph.copy()
        if rounds % 32 == 0:
            bh[0] = 0xFF
        else:
            bh[int(rng.integers(0, bh.size))] ^= 1 << int(rng.integers(0, 8))
        h.decode_start(bh)
        got = np.concatenate([h.decode_bins(cx[i:i + 7]) for i in range(0, n, 7)])
        wantb, wsth = orc.decode_stream(bh, cx, init)
        assert (got == wantb).all() and h.decode_finish() == wsth, "per-bin handle"
        h.close()
    rounds += 1
print("symbol soak ok: %d rounds, %d corrupted streams flagged, %.0f s" % (rounds, flagged, time.time() - t0))
