#!/usr/bin/env python3
"""Randomised soak of the one-process multi-engine path (isscabac_multi: 1..4 engines, all on the test box's one GPU)
against the oracle; test infrastructure, run by hand on a GPU box:   python tests/soak/gpu_fuzz_multi.py [seconds=120] [seed=1]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch  # noqa: F401
from isscabac_amd.engine import MultiEngine, stream_bound
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
    form = int(rng.integers(0, 3))  # 0: bins as bytes, 1: bins as bits, 2: device-resident shards gathered on / pulled from the root
    pitch = max(16, (B + 15) // 16 * 16)
    if form == 1:
        padded = np.zeros((S, pitch), np.uint8)
        padded[:, :B] = bins
        got_p, got_o = eng.encode_packed(np.packbits(padded, axis=1, bitorder="little"), ctx, init, nb)
        dbits, st = eng.decode_packed(got_p, got_o, ctx, init, nb)
        assert not st.any(), ("status packed", S, B, Cn, ep)
        dec = np.unpackbits(dbits, axis=1, bitorder="little")[:, :B]
        for s in range(S):
            n = B if nb is None else int(nb[s])
            assert np.array_equal(dec[s, :n], bins[s, :n]), ("bins packed", S, B, Cn, ep, s)
    elif form == 2:
        dev = torch.device("cuda:0")
        tb, tc, tn, tout, tst = [], [], [], [], []
        bp, cp = np.zeros((S, pitch), np.uint8), np.zeros((S, pitch), np.uint8)
        bp[:, :B], cp[:, :B] = bins, ctx
        for g in range(eng.n):
            lo, hi = eng.shard(S, g)
            tb.append(torch.from_numpy(bp[lo:hi].copy()).to(dev))
            tc.append(torch.from_numpy(cp[lo:hi].copy()).to(dev))
            tn.append(torch.from_numpy((nb if nb is not None else np.full(S, B, np.uint32))[lo:hi].astype(np.int32)).to(dev))
            tout.append(torch.zeros((max(1, hi - lo), pitch), dtype=torch.uint8, device=dev))
            tst.append(torch.ones(max(1, hi - lo), dtype=torch.int32, device=dev))
        cap = S * stream_bound(B)
        dpay = torch.zeros(cap, dtype=torch.uint8, device=dev)
        doff = torch.zeros(S + 1, dtype=torch.int64, device=dev)
        ptr = lambda ts: [t.data_ptr() for t in ts]
        total = eng.encode_dev(ptr(tb), ptr(tc), S, B, pitch, init, dpay.data_ptr(), cap, doff.data_ptr(), d_nbins=ptr(tn))
        assert doff.cpu().numpy().astype(np.uint64).tolist() == want_o.tolist(), ("offsets dev", S, B, Cn, ep)
        assert dpay[:total].cpu().numpy().tobytes() == want_p.tobytes(), ("payload dev", S, B, Cn, ep)
        eng.decode_dev(dpay.data_ptr(), total, doff.data_ptr(), ptr(tc), S, B, pitch, init, ptr(tout), ptr(tst), d_nbins=ptr(tn))
        torch.cuda.synchronize()
        for g in range(eng.n):
            lo, hi = eng.shard(S, g)
            if hi == lo:
                continue
            assert not tst[g].cpu().numpy().any(), ("status dev", S, B, Cn, ep, g)
            d = tout[g].cpu().numpy()
            for s in range(lo, hi):
                n = B if nb is None else int(nb[s])
                assert np.array_equal(d[s - lo, :n], bins[s, :n]), ("bins dev", S, B, Cn, ep, s)
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
print("multi soak (byte, packed and device-resident forms) ok: %d rounds, %.1f M bins, %d corrupted streams flagged, %.0f s" % (rounds, bins_total / 1e6, flagged, time.time() - t0))
