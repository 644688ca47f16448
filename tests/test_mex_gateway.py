"""The MEX gateway (matlab/SimpleCABACMex.cpp) compiled against a stub mex.h: the reference's command strings,
arities, return shapes and error texts (ref: CABAC/SimpleCABACMex.cpp:100-472), plus the batch commands."""
import numpy as np
import pytest

from mex_gateway import Gateway, MexError


@pytest.fixture(scope="module")
def gw():
    return Gateway()


@pytest.fixture(params=["host", pytest.param("gpu", marks=pytest.mark.gpu)])
def perbin(request, monkeypatch):
    """who serves the per-bin commands: the host twin of the handle (the gateway's default; these tests then need no GPU and run
    in the CPU suite) or the device-resident handle (ISSCABAC_MEX_PERBIN=gpu, read by the gateway at init time)"""
    monkeypatch.setenv("ISSCABAC_MEX_PERBIN", request.param)
    return request.param


def test_compiles_and_rejects_like_the_reference(gw):
    with pytest.raises(MexError, match="input 0 must be a valid keyword"):
        gw.call()
    with pytest.raises(MexError, match="input 0 must be a valid keyword"):
        gw.call(np.array([1.0]))
    with pytest.raises(MexError, match="Invalid Command"):
        gw.call("noSuchCommand")
    with pytest.raises(MexError, match="provide two variables to store the trace"):
        gw.call("getEncoderStats", np.array([1.0]), 0.0)  # ref: SimpleCABACMex.cpp:362-365
    with pytest.raises(MexError, match="No initialized CABAC instance"):
        gw.call("getEncoderStats", np.array([12345.0]), 0.0, nlhs=2)  # not a handle this gateway gave out
    with pytest.raises(MexError, match="please provide the filename"):
        gw.call("initByProb")
    with pytest.raises(MexError, match="too many parameters"):
        gw.call("initByProb", "f.bin", np.array([0.5]), 1.0)
    with pytest.raises(MexError, match="invalid filename"):
        gw.call("initByProb", np.array([1.0]), np.array([0.5]))
    with pytest.raises(MexError, match="invalid context initialization"):
        gw.call("initByProb", "f.bin", np.array([1, 2], np.uint8))
    with pytest.raises(MexError, match="pointer to the initialized CABAC engine"):
        gw.call("encodeStart")
    with pytest.raises(MexError, match="No initialized CABAC instance"):
        gw.call("encodeStart", 0.0)


@pytest.mark.gpu
def test_packed_batch_commands_match_the_byte_commands(gw, oracle, tmp_path):
    """'encodeBatchPacked' / 'decodeBatchPacked': the bins as bits across the gateway, same payload and bins as the byte commands"""
    rng = np.random.default_rng(3)
    B, S = 200, 37
    ctx = rng.integers(0, 3, (B, S)).astype(np.uint8)
    bins = rng.integers(0, 2, (B, S)).astype(np.uint8)
    (h,) = gw.call("initByProb", str(tmp_path / "p.bin"), np.array([0.5, 0.3, 0.8]), nlhs=1)
    payload, offsets = gw.call("encodeBatch", h, bins, ctx, nlhs=2)
    pitch = (B + 15) // 16 * 16
    padded = np.zeros((pitch, S), np.uint8)
    padded[:B] = bins
    bits = np.packbits(padded, axis=0, bitorder="little")
    p2, o2 = gw.call("encodeBatchPacked", h, bits, ctx, nlhs=2)
    assert p2.tobytes() == payload.tobytes() and (o2 == offsets).all()
    dbits, status = gw.call("decodeBatchPacked", h, p2, o2, ctx, nlhs=2)
    assert dbits.shape == (pitch // 8, S) and (status == 0).all()
    assert (np.unpackbits(dbits, axis=0, bitorder="little")[:B] == bins).all()
    with pytest.raises(MexError, match="packed bins must have"):
        gw.call("encodeBatchPacked", h, bits[:-1], ctx, nlhs=2)
    # 16-bit context indices have no packed form: both commands refuse a uint16 ctx with a message (B not a multiple of 16 here:
    # the case that once reached a NULL base in the padding copy of decodeBatchPacked), and B = 208 (a multiple of 16) as well
    for rows in (B, 208):
        c16 = np.zeros((rows, S), np.uint16)
        with pytest.raises(MexError, match="no packed form for 16-bit context indices"):
            gw.call("encodeBatchPacked", h, np.zeros(((rows + 15) // 16 * 2, S), np.uint8), c16, nlhs=2)
        with pytest.raises(MexError, match="no packed form for 16-bit context indices"):
            gw.call("decodeBatchPacked", h, p2, o2, c16, nlhs=2)


def test_per_bin_commands_round_trip_and_numbits(gw, perbin, oracle, golden, tmp_path):
    s = {x["name"]: x for x in golden["streams"]}["lcg200_3ctx"]
    fn = str(tmp_path / "mex.bin")
    (h,) = gw.call("initByProb", fn, np.array([0.5, 0.5, 0.5]), nlhs=1)
    assert h.shape == (1, 1)
    gw.call("encodeStart", h)
    for b, c in zip(s["bins_a"], s["ctx_a"]):
        gw.call("encodeBin", h, float(b), float(c))
    (bits,) = gw.call("getNumBits", h, nlhs=1)
    assert bits[0, 0] == 184  # SURVEY 8(c) known answer 3
    with pytest.raises(MexError, match="should either be 1 or 0"):
        gw.call("encodeBin", h, 2.0, 0.0)
    with pytest.raises(MexError, match="provide the bin and context index"):
        gw.call("encodeBin", h, 1.0)
    gw.call("encodeFinish", h)
    assert open(fn, "rb").read().hex() == s["payload"]
    (h2,) = gw.call("initByProb", fn, np.array([0.5, 0.5, 0.5]), nlhs=1)
    gw.call("decodeStart", h2)
    with pytest.raises(MexError, match="provide a variable to store the decoded bin"):
        gw.call("decodeBin", h2, 0.0)
    got = [int(gw.call("decodeBin", h2, float(c), nlhs=1)[0][0, 0]) for c in s["ctx_a"]]
    assert got == s["bins_a"].tolist()
    gw.call("decodeFinish", h2)
    # initByState: flat [ctxIdx mps state] triples, ctxIdx ignored (ref: CABAC_ContextModelsInit.cpp:72-74)
    fn2 = str(tmp_path / "mex2.bin")
    (h3,) = gw.call("initByState", fn2, np.array([9, 0, 20, 9, 1, 0], np.float64), nlhs=1)
    gw.call("encodeStart", h3)
    for b, c in zip([0, 0, 1, 0, 1, 1, 1, 1, 0, 1, 1, 1], [0] * 6 + [1] * 6):
        gw.call("encodeBin", h3, float(b), float(c))
    gw.call("encodeFinish", h3)
    want = oracle.encode_stream([0, 0, 1, 0, 1, 1, 1, 1, 0, 1, 1, 1], [0] * 6 + [1] * 6, [40, 1])
    assert open(fn2, "rb").read() == want.tobytes()
    for hh in (h, h2, h3):
        gw.call("delete", hh)
    with pytest.raises(MexError, match="bitstreamfile access error"):
        (h4,) = gw.call("initByProb", str(tmp_path / "no_such_dir" / "x.bin"), np.array([0.5]), nlhs=1)
        gw.call("encodeStart", h4)


@pytest.mark.gpu
def test_batch_commands(gw, oracle, tmp_path):
    S, B, C = 70, 500, 23  # B not a multiple of 16: the gateway pads the row pitch
    bins, ctx = oracle.synth(0xCABAC003, 0, S, B, C)
    p0 = (np.arange(C) + 1) / (C + 1.0)
    (h,) = gw.call("initByProb", str(tmp_path / "b.bin"), p0, nlhs=1)
    payload, offsets = gw.call("encodeBatch", h, bins.T.copy(), ctx.T.copy(), nlhs=2)  # MATLAB: B x S, one stream per column
    init = oracle.init_by_prob(p0)
    want_payload, want_off = oracle.encode_batch(bins, ctx, init)
    assert payload.shape == (want_payload.size, 1) and payload.ravel().tobytes() == want_payload.tobytes()
    assert offsets.ravel().tolist() == want_off.tolist()
    dec, status = gw.call("decodeBatch", h, payload, offsets, ctx.T.copy(), nlhs=2)
    assert dec.shape == (B, S) and (dec.T == bins).all() and (status == 0).all()
    nb = np.arange(S, dtype=np.uint32) * 7 % (B + 1)
    payload, offsets = gw.call("encodeBatch", h, bins.T.copy(), ctx.T.copy(), nb, nlhs=2)
    want_payload, want_off = oracle.encode_batch(bins, ctx, init, nb)
    assert payload.ravel().tobytes() == want_payload.tobytes() and offsets.ravel().tolist() == want_off.tolist()
    with pytest.raises(MexError, match="same size"):
        gw.call("encodeBatch", h, bins.T.copy(), ctx.T[:10].copy(), nlhs=2)


def test_context_index_is_range_checked_before_narrowing(gw, perbin, tmp_path):
    """256 must not become context 0, and -1 / 255 / 511 must not become a bypass bin (round-1 finding)"""
    (h,) = gw.call("initByProb", str(tmp_path / "r.bin"), np.array([0.5, 0.5, 0.5]), nlhs=1)
    gw.call("encodeStart", h)
    for bad in (3.0, 255.0, 256.0, 511.0, -1.0, 0.5, 1e9, float("nan")):
        with pytest.raises(MexError, match="invalid context index"):
            gw.call("encodeBin", h, 1.0, bad)
    gw.call("encodeBin", h, 1.0, 2.0)
    gw.call("encodeFinish", h)
    gw.call("decodeStart", h)
    for bad in (3.0, 255.0, 256.0, -1.0):
        with pytest.raises(MexError, match="invalid context index"):
            gw.call("decodeBin", h, bad, nlhs=1)
    assert gw.call("decodeBin", h, 2.0, nlhs=1)[0][0, 0] == 1
    (st,) = gw.call("decodeFinish", h, nlhs=1)
    assert st[0, 0] == 0
    gw.call("delete", h)
    with pytest.raises(MexError, match="No initialized CABAC instance"):
        gw.call("encodeStart", h)  # a deleted handle is refused, not dereferenced


def test_handles_hold_the_mex_file_and_are_closed_at_exit(gw, perbin, tmp_path):
    gw.shutdown()  # whatever earlier tests left open
    assert gw.locks() == 0
    hs = [gw.call("initByProb", str(tmp_path / ("l%d.bin" % i)), np.array([0.5]), nlhs=1)[0] for i in range(3)]
    assert gw.locks() == 1  # locked while handles live: `clear mex` cannot unload the engine under them
    gw.call("delete", hs[0])
    assert gw.locks() == 1
    gw.call("delete", hs[1])
    gw.call("delete", hs[2])
    assert gw.locks() == 0
    (h,) = gw.call("initByProb", str(tmp_path / "l.bin"), np.array([0.5]), nlhs=1)
    gw.shutdown()  # MATLAB exit: the gateway closes the stream before destroying the engine
    assert gw.locks() == 0
    with pytest.raises(MexError, match="No initialized CABAC instance"):
        gw.call("encodeStart", h)
    (h,) = gw.call("initByProb", str(tmp_path / "l.bin"), np.array([0.5]), nlhs=1)  # and comes up again
    gw.call("delete", h)


@pytest.mark.gpu
def test_more_than_255_contexts_per_bin_commands_only(gw, oracle, tmp_path):
    """the reference holds up to 999 context models per handle (CommonDef.h:56): the per-bin commands reach all of them"""
    C = 999
    p0 = (np.arange(C) % 200 + 20) / 255.0
    fn = str(tmp_path / "big.bin")
    (h,) = gw.call("initByProb", fn, p0, nlhs=1)
    rng = np.random.default_rng(3)
    ctx = np.concatenate([[0, 255, 256, 511, 998], rng.integers(0, C, 300)])
    bins = rng.integers(0, 2, ctx.size)
    gw.call("encodeStart", h)
    for b, c in zip(bins, ctx):
        gw.call("encodeBin", h, float(b), float(c))
    gw.call("encodeFinish", h)
    want, _ = oracle.encode_ops([0] * ctx.size, bins.tolist(), ctx.tolist(), oracle.init_by_prob(p0))
    assert open(fn, "rb").read() == want.tobytes()
    with pytest.raises(MexError, match="invalid context index"):
        gw.call("encodeBin", h, 1.0, 999.0)
    with pytest.raises(MexError, match="at most 255 contexts"):
        gw.call("encodeBatch", h, np.zeros((16, 1), np.uint8), np.zeros((16, 1), np.uint8), nlhs=2)
    # ... and so do the batch commands once ctx is a uint16 matrix (65535 = bypass bin): isscabac_*_batch16
    S, B = 40, 210
    c16 = rng.integers(0, C, (B, S)).astype(np.uint16)
    c16[rng.random((B, S)) < 0.1] = 0xFFFF
    c16[:5, 0] = [0, 255, 256, 511, 998]
    bb = rng.integers(0, 2, (B, S)).astype(np.uint8)
    payload, offsets = gw.call("encodeBatch", h, bb, c16, nlhs=2)
    init = oracle.init_by_prob(p0)
    offsets = offsets.ravel().astype(np.uint64)
    for s in range(S):
        kind = (c16[:, s] == 0xFFFF).astype(np.uint8)
        want, _ = oracle.encode_ops(kind.tolist(), bb[:, s].tolist(), np.where(kind == 1, 0, c16[:, s]).tolist(), init)
        assert payload.ravel()[int(offsets[s]):int(offsets[s + 1])].tobytes() == want.tobytes(), s
    dec, status = gw.call("decodeBatch", h, payload, offsets.astype(np.float64), c16, nlhs=2)
    assert (dec == bb).all() and (status == 0).all()
    with pytest.raises(MexError, match="invalid context initialization"):
        gw.call("initByProb", fn, np.full(1000, 0.5), nlhs=1)
    gw.call("delete", h)


def test_stats_ep_and_probability_commands(gw, perbin, oracle, tmp_path):
    rng = np.random.default_rng(5)
    fn = str(tmp_path / "s.bin")
    p0 = np.array([0.2, 0.5, 0.8])
    init = oracle.init_by_prob(p0)
    (h,) = gw.call("initByProb", fn, p0, nlhs=1)
    gw.call("encodeStart", h)
    kind, val, arg, bins, ctx = [], [], [], [], []
    for _ in range(400):
        r = rng.integers(0, 4)
        b = int(rng.integers(0, 2))
        if r < 2:
            c = int(rng.integers(0, 3))
            gw.call("encodeBin", h, float(b), float(c))
            kind.append(0); arg.append(c); bins.append(b); ctx.append(c)
        elif r == 2:
            gw.call("encodeBinEP", h, float(b))
            kind.append(1); arg.append(0)
        else:
            p = int(rng.choice([1, 10, 30, 45, 50, 55, 90, 99]))
            gw.call("encodeBinProb", h, float(b), float(p))
            kind.append(3); arg.append(p)
        val.append(b)
    gw.call("encodeFinish", h)
    want, _ = oracle.encode_ops(kind, val, arg, init)
    assert open(fn, "rb").read() == want.tobytes()
    for bad in (46.0, 54.0, 0.0, 100.0, 12.5):
        with pytest.raises(MexError):
            gw.call("encodeBinProb", h, 1.0, bad)
    gw.call("decodeStart", h)
    got = []
    for k, a in zip(kind, arg):
        if k == 0:
            got.append(gw.call("decodeBin", h, float(a), nlhs=1)[0][0, 0])
        elif k == 1:
            got.append(gw.call("decodeBinEP", h, nlhs=1)[0][0, 0])
        else:
            got.append(gw.call("decodeBinProb", h, float(a), nlhs=1)[0][0, 0])
    assert got == val
    gw.call("decodeFinish", h)
    for c in range(3):
        for cmdname, side in (("getEncoderStats", False), ("getDecoderStats", True)):
            trace, stats = gw.call(cmdname, h, float(c), nlhs=2)
            want_t, want_x = oracle.trace_stats(bins, ctx, init, c, decoder=side)
            assert trace.shape == (5, want_t.shape[0]) and (trace.T == want_t).all()
            # the reference fills its 128 x 128 output row-major ([state before][state after], SimpleCABACMex.cpp:393-399):
            # MATLAB therefore sees the transpose
            assert stats.shape == (128, 128) and stats.dtype == np.uint32 and (stats.T == want_x).all()
    with pytest.raises(MexError, match="invalid context index"):
        gw.call("getEncoderStats", h, 3.0, nlhs=2)
    # a damaged stream: the reference asserts in decodeFinish (compiled out in a release MEX); here a warning + the flags
    data = bytearray(open(fn, "rb").read())
    data[-1] ^= 0xFF
    open(fn, "wb").write(bytes(data))
    (h2,) = gw.call("initByProb", fn, p0, nlhs=1)
    gw.call("decodeStart", h2)
    gw.printed()
    (st,) = gw.call("decodeFinish", h2, nlhs=1)
    assert st[0, 0] != 0 and "not terminated properly" in gw.printed()
    gw.call("delete", h)
    gw.call("delete", h2)


def _iss_params(nq, rows, method=1, k=0, types=15, nlbp=3):
    return np.array([method, k, nq, 1, types, nlbp, rows], np.float64)


@pytest.mark.gpu
@pytest.mark.parametrize("serial", ["host", "gpu"])
def test_symbol_commands_run_the_iss_coder_in_one_call_per_matrix(gw, oracle, tmp_path, monkeypatch, serial):
    """the gateway calls matlab/+coder/cabacEncode.m / cabacDecode.m make, against the oracle's walk of the reference's
    loops (ISS/+coder/cabacEncode.m:45-70, cabacDecode.m:29-55) on ISS.m's two matrix shapes.  The one stream per matrix is
    coded by the host twin (the gateway's default for up to four whole-matrix streams) or on the device (ISSCABAC_MEX_PERBIN=gpu);
    the context initialisation -- a histogram -- runs on the GPU either way."""
    monkeypatch.setenv("ISSCABAC_MEX_PERBIN", serial)
    rng = np.random.default_rng(6)
    for rows, cols, nq in ((400, 20, 8), (109, 20, 8), (33, 7, 16)):
        x = np.abs(rng.normal(size=(rows, cols)))
        for r in range(1, rows):
            x[r] = 0.7 * x[r - 1] + 0.3 * x[r]
        G = np.minimum((x / np.quantile(x, 0.99) * nq).astype(np.int64), nq - 1).astype(np.uint8)
        # 1. empirical context initialisation + its uint8 form (cabacEncode.m:23-31)
        p0, p0b = gw.call("issInitContextModel", G, _iss_params(nq, 0), nlhs=2)
        want_p0, want_b = oracle.iss_init_context_model(G, nq, "EG", 0, 15, 3)
        assert p0.shape == (1, 23) and np.array_equal(p0.ravel(), want_p0)
        assert p0b.dtype == np.uint8 and (p0b.ravel() == want_b).all()
        # 2. the whole matrix as one stream (what +coder/cabacEncode.m hands over: G(:) and colRows = rows)
        ctx_init = p0b.ravel().astype(np.float64) / 255.0
        fn = str(tmp_path / "iss.bin")
        (h,) = gw.call("initByProb", fn, ctx_init, nlhs=1)
        params = _iss_params(nq, rows)
        one = G.T.reshape(-1, 1).copy()  # MATLAB's G(:)
        payload, offsets, nbins, hist, cost, hbits = gw.call("encodeSymbolsStats", h, one, params, nlhs=6)
        states = oracle.init_by_prob(ctx_init)
        want, want_hist, want_cost, want_H = oracle.iss_encode_matrix_stats(G, nq, "EG", 0, 15, 3, states)
        assert payload.ravel().tobytes() == want.tobytes() and offsets.ravel().tolist() == [0, want.size]
        assert int(nbins[0, 0]) == int(want_hist.sum())
        assert (hist[:23, 0] == want_hist[:23]).all() and (cost[:23, 0] == want_cost[:23]).all()
        assert (hbits.ravel().reshape(cols, rows).T == want_H).all()
        (p2, o2, nb2) = gw.call("encodeSymbols", h, one, params, nlhs=3)
        assert p2.ravel().tobytes() == want.tobytes() and nb2[0, 0] == nbins[0, 0]
        # 3. decode: one call per matrix
        sym, status = gw.call("decodeSymbols", h, payload, offsets, float(rows * cols), params, nlhs=2)
        assert status[0, 0] == 0 and (sym.ravel().reshape(cols, rows).T == G).all()
        gw.call("delete", h)
    with pytest.raises(MexError, match="params must be"):
        gw.call("issInitContextModel", G, np.array([1.0, 0.0]), nlhs=2)


@pytest.mark.gpu
def test_stream_file_commands_and_multi_device_batches(gw, oracle, reference, tmp_path):
    S, B, C = 40, 304, 9
    bins, ctx = oracle.synth(0xCABAC005, 11, S, B, C, True)
    p0 = (np.arange(C) + 1) / (C + 1.0)
    init = oracle.init_by_prob(p0)
    (h,) = gw.call("initByProb", str(tmp_path / "b.bin"), p0, nlhs=1)
    payload, offsets = gw.call("encodeBatch", h, bins.T.copy(), ctx.T.copy(), nlhs=2)
    want_p, want_o = oracle.encode_batch(bins, ctx, init)
    assert payload.ravel().tobytes() == want_p.tobytes()
    # one file per stream, decodable by the REAL reference classes (oracle/_ref)
    prefix = str(tmp_path / "str_")
    gw.call("writeStreamFiles", payload, offsets, prefix)
    for s in (0, 7, S - 1):
        data = np.fromfile(prefix + "%06d.bin" % s, np.uint8)
        got, st, _ = reference.decode_stream(data, ctx[s], init)
        assert st == 0 and (got == bins[s]).all()
    # ... and files the reference's encoder wrote come back as the same batch
    for s in range(S):
        reference.encode_stream(bins[s], ctx[s], init)[0].tofile(prefix + "%06d.bin" % s)
    p2, o2 = gw.call("readStreamFiles", prefix, float(S), nlhs=2)
    assert p2.ravel().tobytes() == want_p.tobytes() and o2.ravel().tolist() == want_o.tolist()
    with pytest.raises(MexError, match="bitstreamfile access error"):
        gw.call("readStreamFiles", str(tmp_path / "nothing_"), 3.0, nlhs=2)
    # two engines behind the gateway (the box has one GPU: both on device 0 -- the sharding and gather code is the same)
    gw.call("setDevices", np.array([0.0, 0.0]))
    p3, o3 = gw.call("encodeBatch", h, bins.T.copy(), ctx.T.copy(), nlhs=2)
    assert p3.ravel().tobytes() == want_p.tobytes() and o3.ravel().tolist() == want_o.tolist()
    dec, status = gw.call("decodeBatch", h, p3, o3, ctx.T.copy(), nlhs=2)
    assert (dec.T == bins).all() and (status == 0).all()
    # ... and the packed and the symbol-level commands under 'setDevices' (three engines: an uneven split)
    gw.call("setDevices", np.array([0.0, 0.0, 0.0]))
    Bm, Sm = bins.shape[1], bins.shape[0]
    pitch = (Bm + 15) // 16 * 16
    padded = np.zeros((pitch, Sm), np.uint8)
    padded[:Bm] = bins.T
    bits = np.packbits(padded, axis=0, bitorder="little")
    p5, o5 = gw.call("encodeBatchPacked", h, bits, ctx.T.copy(), nlhs=2)
    assert p5.ravel().tobytes() == want_p.tobytes() and o5.ravel().tolist() == want_o.tolist()
    dbits, status = gw.call("decodeBatchPacked", h, p5, o5, ctx.T.copy(), nlhs=2)
    assert (status == 0).all() and (np.unpackbits(dbits, axis=0, bitorder="little")[:Bm] == bins.T).all()
    rng = np.random.default_rng(12)
    rows, cols, nq = 50, 11, 8
    G = np.minimum(np.abs(rng.normal(0, 2.0, (rows, cols))).astype(np.int64), nq - 1).astype(np.uint8)
    (hs,) = gw.call("initByProb", str(tmp_path / "sym.bin"), np.full(23, 0.5), nlhs=1)
    params = _iss_params(nq, 0)  # every column a stream
    ps, os_, nbs, hist, cost, hbits = gw.call("encodeSymbolsStats", hs, G, params, nlhs=6)
    init = oracle.init_by_prob(np.full(23, 0.5))
    for c in range(cols):
        want, want_nb = oracle.iss_encode_column(G[:, c], nq, "EG", 0, 15, 3, init)
        assert ps.ravel()[int(os_.ravel()[c]):int(os_.ravel()[c + 1])].tobytes() == want.tobytes() and int(nbs.ravel()[c]) == want_nb
        assert int(hist[:, c].sum()) == want_nb and int(hbits[:, c].sum()) <= 8 * want.size
    sym, status = gw.call("decodeSymbols", hs, ps, os_, float(rows), params, nlhs=2)
    assert (status == 0).all() and (sym == G).all()
    gw.call("delete", hs)
    gw.call("setDevices", np.zeros((0, 0)))
    p4, _ = gw.call("encodeBatch", h, bins.T.copy(), ctx.T.copy(), nlhs=2)
    assert p4.ravel().tobytes() == want_p.tobytes()
    gw.call("delete", h)
