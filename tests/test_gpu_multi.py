"""isscabac_multi: several devices behind one handle in one process.  The GPU box has one GPU, so
the engines of a handle share device 0 -- the shard arithmetic, the per-device host threads and the peer-copy gather are
the code a real multi-GPU node runs."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("ndev", [1, 2, 3])
def test_host_batches_equal_a_single_engine(oracle, ndev):
    from isscabac_amd import MultiEngine
    m = MultiEngine([0] * ndev)
    assert [m.peer_access(g) for g in range(ndev)] == [0] * ndev and m.peer_access(ndev) == -1  # one device here: no pair to ask about
    rng = np.random.default_rng(ndev)
    for S, B, C in ((1, 64, 3), (7, 200, 5), (257, 1040, 64)):
        bins, ctx = oracle.synth(0xCABAC005, 5, S, B, C, True)
        init = rng.integers(0, 126, C).astype(np.uint8)
        nb = rng.integers(0, B + 1, S).astype(np.uint32)
        edges = [m.shard(S, g) for g in range(ndev)]
        assert edges[0][0] == 0 and edges[-1][1] == S and all(a[1] == b[0] for a, b in zip(edges, edges[1:]))
        for nbins in (None, nb):
            want_p, want_o = oracle.encode_batch(bins, ctx, init, nbins)
            got_p, got_o = m.encode(bins, ctx, init, nbins)
            assert (got_o == want_o).all() and got_p.tobytes() == want_p.tobytes()
            dec, status = m.decode(want_p, want_o, ctx, init, nbins)
            mask = np.ones((S, B), bool) if nbins is None else np.arange(B)[None, :] < nbins[:, None]
            assert (status == 0).all() and (dec * mask == bins * mask).all()
    m.close()


@pytest.mark.parametrize("ndev", [2, 4])
def test_device_resident_shards_gathered_on_the_root(oracle, ndev):
    import torch
    from isscabac_amd import MultiEngine
    dev = torch.device("cuda:0")
    m = MultiEngine([0] * ndev)
    S, B, C = 1003, 512, 64  # not divisible by the number of devices
    bins, ctx = oracle.synth(0xCABAC003, 0, S, B, C)
    init = np.zeros(C, np.uint8)
    want_p, want_o = oracle.encode_batch(bins, ctx, init)
    tb, tc = [], []
    for g in range(ndev):
        lo, hi = m.shard(S, g)
        tb.append(torch.from_numpy(bins[lo:hi].copy()).to(dev))
        tc.append(torch.from_numpy(ctx[lo:hi].copy()).to(dev))
    cap = S * (B // 4 + 64)
    payload = torch.zeros(cap, dtype=torch.uint8, device=dev)
    offsets = torch.zeros(S + 1, dtype=torch.int64, device=dev)
    for _ in range(2):  # twice: the staging buffers are reused
        total = m.encode_dev([t.data_ptr() for t in tb], [t.data_ptr() for t in tc], S, B, B, init, payload.data_ptr(), cap,
                             offsets.data_ptr())
        torch.cuda.synchronize()
        assert total == want_p.size
        assert offsets.cpu().numpy().astype(np.uint64).tolist() == want_o.tolist()
        assert payload[:total].cpu().numpy().tobytes() == want_p.tobytes()
        payload.zero_()
        offsets.zero_()
    from isscabac_amd import EngineError
    with pytest.raises(EngineError):
        m.encode_dev([t.data_ptr() for t in tb], [t.data_ptr() for t in tc], S, B, B, init, payload.data_ptr(), 100,
                     offsets.data_ptr())
    m.close()


@pytest.mark.parametrize("ndev", [1, 3, 4])
def test_packed_and_symbol_batches_equal_a_single_engine(oracle, ndev):
    """the forms the one-process path lacked last round: bins as bits, and the symbol-level calls (config 4's path,
    ref: ISS/+coder/cabacEncode.m:45-70) -- split over the engines, results those of one engine / the oracle"""
    from isscabac_amd import MultiEngine, SymbolParams
    m = MultiEngine([0] * ndev)
    rng = np.random.default_rng(20 + ndev)
    S, B, C = 77, 300, 9
    bins, ctx = oracle.synth(0xCABAC005, 3, S, B, C, True)
    init = rng.integers(0, 126, C).astype(np.uint8)
    nb = rng.integers(0, B + 1, S).astype(np.uint32)
    pitch = (B + 15) // 16 * 16
    padded = np.zeros((S, pitch), np.uint8)
    padded[:, :B] = bins
    bits = np.packbits(padded, axis=1, bitorder="little")
    for nbins in (None, nb):
        want_p, want_o = oracle.encode_batch(bins, ctx, init, nbins)
        got_p, got_o = m.encode_packed(bits, ctx, init, nbins)
        assert (got_o == want_o).all() and got_p.tobytes() == want_p.tobytes()
        dbits, status = m.decode_packed(want_p, want_o, ctx, init, nbins)
        dec = np.unpackbits(dbits, axis=1, bitorder="little")[:, :B]
        mask = np.ones((S, B), bool) if nbins is None else np.arange(B)[None, :] < nbins[:, None]
        assert (status == 0).all() and (dec * mask == bins * mask).all()
    # symbols: columns as streams (ragged) and whole matrices as streams
    nq, rows, cols = 8, 40, 13
    G = np.minimum(np.abs(rng.normal(0, 2.0, (rows, cols))).astype(np.int64), nq - 1).astype(np.uint8)
    init = rng.integers(0, 126, 23).astype(np.uint8)
    params = SymbolParams.iss(nq, "EG", 0, 15, 3)
    ns = rng.integers(0, rows + 1, cols).astype(np.uint32)
    ns[:2] = [0, rows]
    payload, offsets, nbins, hist, cost, hb = m.encode_symbols(G.T.copy(), params, init, ns, stats=True)
    for c in range(cols):
        want, want_nb = oracle.iss_encode_column(G[:ns[c], c], nq, "EG", 0, 15, 3, init)
        assert payload[int(offsets[c]):int(offsets[c + 1])].tobytes() == want.tobytes() and int(nbins[c]) == want_nb
        assert int(hist[c].sum()) == want_nb
    sym, status = m.decode_symbols(payload, offsets, rows, params, init, ns)
    assert (status == 0).all() and all((sym[c, :ns[c]] == G[:ns[c], c]).all() for c in range(cols))
    mats = [np.minimum(np.abs(rng.normal(0, 1.5, (rows, 4))).astype(np.int64), nq - 1).astype(np.uint8) for _ in range(5)]
    params = SymbolParams.iss(nq, "EG", 0, 15, 3, col_rows=rows)
    one = np.stack([g.T.reshape(-1) for g in mats])
    payload, offsets, nbins = m.encode_symbols(one, params, init)
    for i, g in enumerate(mats):
        want, want_nb = oracle.iss_encode_matrix(g, nq, "EG", 0, 15, 3, init)
        assert payload[int(offsets[i]):int(offsets[i + 1])].tobytes() == want.tobytes() and int(nbins[i]) == want_nb
    sym, status = m.decode_symbols(payload, offsets, rows * 4, params, init)
    assert (status == 0).all() and (sym == one).all()
    m.close()


@pytest.mark.parametrize("ndev", [1, 2, 3])
def test_device_resident_round_trip_through_the_root(oracle, ndev):
    """encode_batch_dev leaves payload + offsets on the root device; decode_batch_dev pulls every device's slice back out
    of them and decodes into that device's own buffers (ragged lengths, a stream count the devices do not divide)"""
    import torch
    from isscabac_amd import MultiEngine
    dev = torch.device("cuda:0")
    m = MultiEngine([0] * ndev)
    S, B, C = 1003, 512, 64
    bins, ctx = oracle.synth(0xCABAC005, 11, S, B, C, True)
    rng = np.random.default_rng(5)
    init = rng.integers(0, 126, C).astype(np.uint8)
    nb = rng.integers(0, B + 1, S).astype(np.uint32)
    want_p, want_o = oracle.encode_batch(bins, ctx, init, nb)
    tb, tc, tn, tout, tst = [], [], [], [], []
    for g in range(ndev):
        lo, hi = m.shard(S, g)
        tb.append(torch.from_numpy(bins[lo:hi].copy()).to(dev))
        tc.append(torch.from_numpy(ctx[lo:hi].copy()).to(dev))
        tn.append(torch.from_numpy(nb[lo:hi].astype(np.int32)).to(dev))
        tout.append(torch.zeros((hi - lo, B), dtype=torch.uint8, device=dev))
        tst.append(torch.ones(hi - lo, dtype=torch.int32, device=dev))
    cap = S * (B // 4 + 64)
    payload = torch.zeros(cap, dtype=torch.uint8, device=dev)
    offsets = torch.zeros(S + 1, dtype=torch.int64, device=dev)
    total = m.encode_dev([t.data_ptr() for t in tb], [t.data_ptr() for t in tc], S, B, B, init, payload.data_ptr(), cap,
                         offsets.data_ptr(), d_nbins=[t.data_ptr() for t in tn])
    assert total == want_p.size and payload[:total].cpu().numpy().tobytes() == want_p.tobytes()
    for _ in range(2):  # twice: the staging buffers are reused
        m.decode_dev(payload.data_ptr(), total, offsets.data_ptr(), [t.data_ptr() for t in tc], S, B, B, init,
                     [t.data_ptr() for t in tout], [t.data_ptr() for t in tst], d_nbins=[t.data_ptr() for t in tn])
        torch.cuda.synchronize()
        for g in range(ndev):
            lo, hi = m.shard(S, g)
            mask = np.arange(B)[None, :] < nb[lo:hi, None]
            assert (tst[g].cpu().numpy() == 0).all()
            assert (tout[g].cpu().numpy() * mask == bins[lo:hi] * mask).all()
            tout[g].zero_()
            tst[g].fill_(1)
    from isscabac_amd import EngineError
    offsets[5] = 10 ** 12  # a boundary beyond the payload: refused before any device reads
    if ndev > 1:
        lo1, _ = m.shard(S, 1)
        offsets[lo1] = 10 ** 12
        with pytest.raises(EngineError):
            m.decode_dev(payload.data_ptr(), total, offsets.data_ptr(), [t.data_ptr() for t in tc], S, B, B, init,
                         [t.data_ptr() for t in tout], [t.data_ptr() for t in tst])
    m.close()


def test_create_multi_leaves_the_callers_device_current():
    """isscabac_create() switches the HIP device; the multi entry points must hand the caller's device back (both exits)"""
    import torch
    from isscabac_amd import EngineError, MultiEngine
    n = torch.cuda.device_count()
    torch.cuda.set_device(0)
    ids = [0, n - 1] if n > 1 else [0, 0]  # with more than one device: the list ends on a device that is not the current one
    m = MultiEngine(ids)
    assert torch.cuda.current_device() == 0
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    cur = ctypes.c_int(-1)
    assert hip.hipGetDevice(ctypes.byref(cur)) == 0 and cur.value == 0
    m.close()
    with pytest.raises(EngineError):
        MultiEngine([0, 10 ** 6])  # the second create fails: the error exit restores the device too
    assert hip.hipGetDevice(ctypes.byref(cur)) == 0 and cur.value == 0
