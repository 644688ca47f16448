"""Very long streams (up to 3 M bins) against the oracle, including corrupted payloads -- run by hand on a GPU box."""
import sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from isscabac_amd import Engine
from oracle.pyoracle import Oracle
eng, orc = Engine(0), Oracle()
rng = np.random.default_rng(3)
for S, B, C, ep in ((130, 70000, 64, 0.0), (5, 1 << 20, 23, 0.1), (2, 3000000, 255, 0.0), (70, 200000, 3, 0.5)):
    init = rng.integers(0, 126, C).astype(np.uint8)
    ctx = rng.integers(0, C, (S, B)).astype(np.uint8)
    ctx[rng.random((S, B)) < ep] = 0xFF
    skew = rng.random(C + 1) ** 2
    bins = (rng.random((S, B)) < skew[np.where(ctx == 0xFF, C, ctx)]).astype(np.uint8)
    nb = rng.integers(B // 2, B + 1, S).astype(np.uint32)
    wp, wo = orc.encode_batch(bins, ctx, init, nb)
    gp, go = eng.encode(bins, ctx, init, nb)
    assert np.array_equal(go, wo) and gp.tobytes() == wp.tobytes(), (S, B)
    dec, st = eng.decode(gp, go, ctx, init, nb)
    assert not st.any()
    for s in range(S):
        assert np.array_equal(dec[s, :nb[s]], bins[s, :nb[s]])
    bad = gp.copy(); idx = rng.integers(0, bad.size, 50); bad[idx] ^= 0x10
    dec, st = eng.decode(bad, go, ctx, init, nb)
    wd, ws = orc.decode_batch(bad, wo, ctx, init, nb)
    assert np.array_equal(st, ws)
    for s in range(S):
        assert np.array_equal(dec[s, :nb[s]], wd[s, :nb[s]])
    print("ok", S, B, C, ep, "payload", gp.size)
