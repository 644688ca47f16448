"""SURVEY 8(f)-4 on the GPU, through the C ABI: fixed-probability bins, state traces, TRk / FL32 binarisers, and the
999-context single-stream handle.  Expectations: tests/golden/cabac_golden_f4.json (the reference compiled with its two
switches on) and the oracle."""
import json
import os

import numpy as np
import pytest

from conftest import unbits, unhex

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAFE_PROBS = [p for p in range(1, 100) if not (46 <= p <= 49 or 51 <= p <= 54)]


@pytest.fixture(scope="module")
def golden4():
    with open(os.path.join(ROOT, "tests", "golden", "cabac_golden_f4.json")) as f:
        return json.load(f)


def dense(sp):
    t = np.zeros((128, 128), np.uint32)
    for i, j, n in sp:
        t[i, j] = n
    return t


def ops16(kind, arg):
    """golden op scripts -> 16-bit handle ops; a multi-EP op (kind 2) is its bins as single bypass ops, MSB first"""
    from isscabac_amd.engine import OP_BYPASS, op_prob
    return OP_BYPASS if kind == 1 else op_prob(arg) if kind == 3 else int(arg)


def expand(op):
    bins, ops, groups = [], [], []
    for k, v, a in zip(op["kind"], op["val"], op["arg"]):
        if k == 2:
            bins += [(v >> (a - 1 - i)) & 1 for i in range(a)]
            ops += [0xFFFF] * a
            groups.append(a)
        else:
            bins.append(v)
            ops.append(ops16(k, a))
            groups.append(1)
    return np.array(bins, np.uint8), np.array(ops, np.uint16), groups


def test_batch_fixed_probability_streams_match_the_reference(engine, golden4):
    for s in golden4["streams"]:
        bins, ctx, init = unbits(s["bins"], s["n"]), unhex(s["ctx"]), unhex(s["ctx_init"])
        payload, offsets = engine.encode(bins[None, :], ctx[None, :], init)
        assert payload.tobytes().hex() == s["payload"], s["name"]
        dec, status = engine.decode(payload, offsets, ctx[None, :], init)
        assert status[0] == 0 and (dec[0] == bins).all(), s["name"]


def test_batch_fixed_probability_mixed_with_everything_vs_oracle(engine, oracle):
    rng = np.random.default_rng(41)
    for C, S, B in ((64, 700, 528), (128, 300, 1040), (3, 130, 96)):
        init = rng.integers(0, 126, C).astype(np.uint8)
        ctx = rng.integers(0, C, (S, B)).astype(np.uint8)
        bins = rng.integers(0, 2, (S, B)).astype(np.uint8)
        r = rng.random((S, B))
        probs = rng.choice(SAFE_PROBS, (S, B))
        isp = r < 0.3
        ctx[isp] = (155 + probs[isp]).astype(np.uint8)
        bins[isp] = (rng.random(int(isp.sum())) * 100 < probs[isp]).astype(np.uint8)
        ctx[(r >= 0.3) & (r < 0.45)] = 0xFF
        ctx[::7, :64] = rng.integers(0, C, (ctx[::7].shape[0], 64))  # some plain chunks in between
        nb = rng.integers(0, B + 1, S).astype(np.uint32)
        want_p, want_o = oracle.encode_batch(bins, ctx, init, nb)
        got_p, got_o = engine.encode(bins, ctx, init, nb)
        assert (got_o == want_o).all() and got_p.tobytes() == want_p.tobytes(), C
        dec, status = engine.decode(want_p, want_o, ctx, init, nb)
        mask = np.arange(B)[None, :] < nb[:, None]
        assert (status == 0).all() and (dec * mask == bins * mask).all(), C
        # corrupt payloads: the fixed-probability path must produce the reference's garbage and flags too
        bad = want_p.copy()
        bad[rng.integers(0, bad.size, bad.size // 50)] ^= rng.integers(1, 256, bad.size // 50).astype(np.uint8)
        bad[want_o[:-1][::9].astype(np.int64)] = 0xFF  # streams that start with 0xFF take the reference-formulation decoder
        wd, ws = oracle.decode_batch(bad, want_o, ctx, init, nb)
        gd, gs = engine.decode(bad, want_o, ctx, init, nb)
        assert (gs == ws).all() and (gd * mask == wd * mask).all(), C


def test_unsupported_probabilities_leave_other_streams_alone(engine, oracle):
    rng = np.random.default_rng(42)
    S, B, C = 128, 320, 16
    init = rng.integers(0, 126, C).astype(np.uint8)
    ctx = rng.integers(0, C, (S, B)).astype(np.uint8)
    bins = rng.integers(0, 2, (S, B)).astype(np.uint8)
    dirty = ctx.copy()
    dirty[3, ::5] = 155 + 47
    dirty[77, ::3] = 155 + 53
    want_p, want_o = oracle.encode_batch(bins, ctx, init)
    got_p, got_o = engine.encode(bins, dirty, init)
    for s in range(S):
        if s not in (3, 77):
            assert got_p[int(got_o[s]):int(got_o[s + 1])].tobytes() == want_p[int(want_o[s]):int(want_o[s + 1])].tobytes(), s


def test_handle_op_scripts_match_the_reference(engine, golden4):
    from isscabac_amd.engine import Stream
    for op in golden4["ops"]:
        bins, ops, groups = expand(op)
        st = Stream(engine, unhex(op["ctx_init"]))
        st.encode_start()
        st.encode_ops(bins, ops)
        assert st.encode_finish().tobytes().hex() == op["payload"], op["name"]
        st.decode_start(unhex(op["payload"]))
        got = st.decode_ops(ops)
        assert st.decode_finish() == 0 and (got == bins).all(), op["name"]
        st.close()
        # one op per call, the way the MATLAB loops drive the handle (a fresh handle: like the reference's, a handle's
        # decoder-side context models are initialised once, at init)
        st = Stream(engine, unhex(op["ctx_init"]))
        st.encode_start()
        for b, o in zip(bins[:60], ops[:60]):
            st.encode_ops([b], [o])
            assert st.num_bits() % 8 == 0
        part = st.encode_finish()
        st.decode_start(part)
        assert [int(st.decode_ops([o])[0]) for o in ops[:60]] == bins[:60].tolist() and st.decode_finish() == 0
        st.close()


def test_handle_refuses_unsupported_probabilities_and_bad_contexts(engine):
    from isscabac_amd.engine import EngineError, Stream, op_prob
    st = Stream(engine, np.zeros(3, np.uint8))
    st.encode_start()
    for bad in (op_prob(46), op_prob(49), op_prob(51), op_prob(54), op_prob(0), op_prob(100), 3, 998):
        with pytest.raises(EngineError):
            st.encode_ops([1], [bad])
    st.encode_ops([1, 0, 1], [op_prob(45), op_prob(55), op_prob(50)])
    payload = st.encode_finish()
    st.decode_start(payload)
    with pytest.raises(EngineError):
        st.decode_ops([op_prob(47)])
    assert st.decode_ops([op_prob(45), op_prob(55), op_prob(50)]).tolist() == [1, 0, 1] and st.decode_finish() == 0


def test_handle_state_traces_match_the_reference(engine, golden4):
    from isscabac_amd.engine import Stream
    for t in golden4["traces"]:
        bins, ctx, init = unbits(t["bins"], t["n"]), unhex(t["ctx"]), unhex(t["ctx_init"])
        st = Stream(engine, init)
        st.set_trace(True)
        st.encode_start()
        # in pieces: short calls travel in the kernel arguments, long ones through device memory
        cuts = sorted(set([0, 1, 7, 40, 41, t["n"] // 2, t["n"]]))
        for a, b in zip(cuts, cuts[1:]):
            if a < t["n"]:
                st.encode_bins(bins[a:min(b, t["n"])], ctx[a:min(b, t["n"])])
        payload = st.encode_finish()
        assert payload.tobytes().hex() == t["payload"], t["name"]
        st.decode_start(payload)
        got = np.concatenate([st.decode_bins(ctx[a:b]) for a, b in zip(cuts, cuts[1:]) if a < t["n"]]) if t["n"] else bins
        assert (got == bins).all() and st.decode_finish() == 0
        for side, key in ((False, "encoder"), (True, "decoder")):
            for c in range(t["nctx"]):
                trace, trans = st.get_stats(c, decoder=side)
                assert trace.tobytes().hex() == t[key][c]["trace"], (t["name"], key, c)
                assert (trans == dense(t[key][c]["transitions"])).all(), (t["name"], key, c)
        st.close()


def test_handle_with_999_contexts(engine, oracle):
    """the reference's handle holds up to 999 context models (CommonDef.h:56): every one of them is reachable through the
    16-bit ops, and the batch calls (one byte per bin) refuse more than 255 instead of aliasing them"""
    from isscabac_amd.engine import EngineError, Stream
    rng = np.random.default_rng(43)
    C, n = 999, 6000
    init = rng.integers(0, 126, C).astype(np.uint8)
    ctx = rng.integers(0, C, n)
    ctx[:C] = rng.permutation(C)  # every context at least once
    bins = (rng.random(n) < rng.random(C)[ctx]).astype(np.uint8)
    want, fin = oracle.encode_ops([0] * n, bins.tolist(), ctx.tolist(), init)
    st = Stream(engine, init)
    st.set_trace(True)
    st.encode_start()
    st.encode_ops(bins, ctx.astype(np.uint16))
    assert st.encode_finish().tobytes() == want.tobytes()
    st.decode_start(want)
    assert (st.decode_ops(ctx.astype(np.uint16)) == bins).all() and st.decode_finish() == 0
    for c in (0, 255, 256, 511, 998):
        trace, trans = st.get_stats(c)
        wt, wx = oracle_trace_999(oracle, bins, ctx, init, c)
        assert trace.tobytes() == wt.tobytes() and (trans == wx).all()
    with pytest.raises(EngineError):
        Stream(engine, np.zeros(1000, np.uint8))
    with pytest.raises(EngineError):  # batch calls: a documented, tested error
        engine.encode(np.zeros((1, 16), np.uint8), np.zeros((1, 16), np.uint8), np.zeros(256, np.uint8))


def oracle_trace_999(oracle, bins, ctx, init, c):
    """the oracle's trace reduction takes byte contexts: follow context c alone (its state depends on its own bins only)"""
    sel = ctx == c
    return oracle.trace_stats(bins[sel], np.zeros(int(sel.sum()), np.uint8), init[c:c + 1], 0)


def test_symbol_kernel_bin_trace_gives_the_handles_statistics(engine, oracle):
    """the state trace out of the symbol kernels (whole-matrix ISS stream) == the reference's per-bin bookkeeping on the
    bin / context sequence the oracle's matrix walk produces"""
    from isscabac_amd.engine import SymbolParams, trace_stats
    rng = np.random.default_rng(44)
    rows, cols, nq = 60, 9, 8
    G = np.minimum(np.abs(rng.normal(0, 2.0, (rows, cols))).astype(np.int64), nq - 1).astype(np.uint8)
    init = rng.integers(0, 126, 23).astype(np.uint8)
    params = SymbolParams.iss(nq, "EG", 0, 15, 3, col_rows=rows)
    sym = G.T.reshape(1, -1).copy()  # column after column
    payload, offsets, nbins, hist, cost, hb, rec = engine.encode_symbols_stats(sym, params, init, trace=True)
    want, want_nb = oracle.iss_encode_matrix(G, nq, "EG", 0, 15, 3, init)
    assert payload.tobytes() == want.tobytes() and int(nbins[0]) == want_nb
    r = rec[0, :want_nb]
    assert ((r >> 11) & 1).all()
    bins, ctx = ((r >> 10) & 1).astype(np.uint8), (r & 0x3FF).astype(np.uint8)
    assert (np.bincount(ctx, minlength=64)[:64] == hist[0]).all()
    assert oracle.encode_stream(bins, ctx, init).tobytes() == want.tobytes()  # the records ARE the coded sequence
    for c in range(23):
        got_t, got_x = trace_stats(r, c)
        want_t, want_x = oracle.trace_stats(bins, ctx, init, c)
        assert got_t.tobytes() == want_t.tobytes() and (got_x == want_x).all(), c


@pytest.mark.parametrize("method,k,nq", [("TR", 0, 12), ("TR", 1, 16), ("TR", 2, 40), ("FL32", 0, 256), ("FL32", 0, 7)])
def test_tr_and_fl32_binarisers_encode_like_the_oracle(engine, oracle, method, k, nq):
    from isscabac_amd.engine import EngineError, SymbolParams
    rng = np.random.default_rng(45)
    S, R = 70, 50
    sym = rng.integers(0, nq, (S, R)).astype(np.uint8)
    sym[0, :] = nq - 1  # the maximum: TR's all-ones suffix
    for types, nlbp in ((15, 3), (31, 2)):
        nctx = 7 * nlbp + 2
        init = rng.integers(0, 126, nctx).astype(np.uint8)
        params = SymbolParams.iss(nq, method, k, types, nlbp)
        payload, offsets, nbins = engine.encode_symbols(sym, params, init)
        for s in range(S):
            want, want_nb = oracle.iss_encode_column(sym[s], nq, method, k, types, nlbp, init)
            assert payload[int(offsets[s]):int(offsets[s + 1])].tobytes() == want.tobytes(), (s, types)
            assert int(nbins[s]) == want_nb
        with pytest.raises(EngineError):  # no end-of-symbol rule in the reference: not decodable there either
            engine.decode_symbols(payload, offsets, R, params, init)


@pytest.mark.parametrize("live", ["1", "0"])
@pytest.mark.parametrize("tracing", [False, True])
def test_handle_short_calls_with_and_without_the_live_kernel(engine, oracle, live, tracing, monkeypatch):
    """decodeBin, and getNumBits behind a few queued encodeBin, are served by the handle's live kernel (one lane polling the
    mailbox; ISSCABAC_STREAM_LIVE=0 = a launch per call): the same bytes, byte counts, bins, traces and status either way.
    Calls of every size are mixed -- short ones go through the live kernel, long ones stop it and launch -- and the handle
    is left alone for longer than the kernel's idle time in between, so that it leaves and is started again."""
    import time
    from isscabac_amd import Stream
    monkeypatch.setenv("ISSCABAC_STREAM_LIVE", live)
    rng = np.random.default_rng(int(live) * 2 + int(tracing))
    nctx, n = 40, 3000
    ctx = rng.integers(0, nctx, n).astype(np.uint8)
    ctx[rng.random(n) < 0.15] = 0xFF
    p = rng.random(nctx)
    bins = np.where(ctx == 0xFF, rng.integers(0, 2, n), rng.random(n) < p[np.minimum(ctx, nctx - 1)]).astype(np.uint8)
    init = rng.integers(0, 126, nctx).astype(np.uint8)
    want = oracle.encode_stream(bins, ctx, init)
    st = Stream(engine, init)
    st.set_trace(tracing)
    st.encode_start()
    i, bits = 0, []
    sizes = [1, 1, 2, 1, 7, 32, 33, 1, 1, 100, 1, 5, 1, 1, 64, 1]
    k = 0
    while i < n:
        m = min(sizes[k % len(sizes)], n - i)
        st.encode_bins(bins[i:i + m], ctx[i:i + m])
        i += m
        bits.append((i, st.num_bits()))
        if k % 23 == 5:
            time.sleep(0.004)  # the live kernel leaves after 200 us without requests
        k += 1
    for upto, b in bits[::7]:  # getNumBits after a prefix == 8 x the bytes the oracle has written for that prefix
        assert b == oracle.encode_stream(bins[:upto], ctx[:upto], init, want_final=True)[1], upto
    payload = st.encode_finish()
    assert payload.tobytes() == want.tobytes()
    st.decode_start(payload)
    got, i, k = [], 0, 0
    while i < n:
        m = min(sizes[(k + 3) % len(sizes)], n - i)
        got.append(st.decode_bins(ctx[i:i + m]))
        i += m
        if k % 29 == 4:
            time.sleep(0.004)
        k += 1
    assert (np.concatenate(got) == bins).all()
    assert st.decode_finish() == 0
    if tracing:
        cc = int(ctx[ctx != 0xFF][0])
        for side in (False, True):
            got_t, got_x = st.get_stats(cc, decoder=side)
            want_t, want_x = oracle.trace_stats(bins, ctx, init, cc, decoder=side)
            assert got_t.tobytes() == want_t.tobytes() and (got_x == want_x).all(), side
    st.close()


def test_a_polling_live_kernel_never_stands_in_front_of_other_work(engine, oracle):
    """The runtime maps streams onto a few hardware queues: a kernel queued behind a handle's polling live kernel would wait until
    that kernel leaves by itself.  So the engine makes its handles' live kernels leave before it enqueues anything else
    (cabac_internal.h: ENGINE_ENTER, cabac_stream.hip: quiesce_others).  Without that, four handles decoding in turn took
    500 us per call and a batch call right behind a decodeBin 1 ms (tools/live_interleave.py); coarse bounds, far from both."""
    import time
    from isscabac_amd import Stream
    rng = np.random.default_rng(77)
    n = 400
    ctx = rng.integers(0, 3, n).astype(np.uint8)
    bins = rng.integers(0, 2, n).astype(np.uint8)
    init = np.zeros(3, np.uint8)
    payload = oracle.encode_stream(bins, ctx, init)
    hs = [Stream(engine, init) for _ in range(5)]
    for h in hs:
        h.decode_start(payload)
        assert h.decode_bins(ctx[:1])[0] == bins[0]  # (the first call of a process loads the kernel: not timed)
    t0 = time.perf_counter()
    for i in range(1, n):
        for h in hs:
            assert h.decode_bins(ctx[i:i + 1])[0] == bins[i]
    per_call = (time.perf_counter() - t0) / ((n - 1) * len(hs))
    for h in hs[1:]:
        assert h.decode_finish() == 0
        h.close()
    bb = rng.integers(0, 2, (64, 256)).astype(np.uint8)
    cc = rng.integers(0, 3, (64, 256)).astype(np.uint8)
    want = engine.encode(bb, cc, init)[0]
    h = Stream(engine, init)
    h.decode_start(payload)
    lat = []
    for i in range(60):
        h.decode_bins(ctx[i:i + 1])  # leaves a live kernel polling
        t1 = time.perf_counter()
        got = engine.encode(bb, cc, init)[0]
        lat.append(time.perf_counter() - t1)
        assert got.tobytes() == want.tobytes()
    h.close()
    hs[0].close()
    assert per_call < 250e-6, "handles decoding in turn: %.0f us per call" % (per_call * 1e6)
    assert float(np.median(lat)) < 600e-6, "a batch call behind a decodeBin: %.0f us" % (np.median(lat) * 1e6)


def test_a_busy_live_kernel_leaves_after_its_residency_bound_and_the_next_call_goes_on(engine, oracle):
    """A caller that never pauses must not keep its polling kernel resident for ever (a kernel of another engine or HIP user queued
    behind it would wait): the kernel leaves after 2 ms whatever the traffic and the next call starts the next one.  5 000
    back-to-back decodeBin calls (>= 10 ms) therefore cross several kernel generations; every bin must still be the oracle's,
    and a SECOND engine's batch call issued in the middle must come back in far less than the backstop (seconds)."""
    import time
    from isscabac_amd import Engine, Stream
    rng = np.random.default_rng(5)
    n = 5000
    ctx = rng.integers(0, 5, n).astype(np.uint8)
    bins = (rng.random(n) < 0.3).astype(np.uint8)
    init = rng.integers(0, 126, 5).astype(np.uint8)
    payload = oracle.encode_stream(bins, ctx, init)
    other = Engine(0)
    bb = rng.integers(0, 2, (64, 64)).astype(np.uint8)
    cc = rng.integers(0, 5, (64, 64)).astype(np.uint8)
    want = oracle.encode_batch(bb, cc, init)[0]
    h = Stream(engine, init)
    h.decode_start(payload)
    worst = 0.0
    for i in range(n):
        assert h.decode_bins(ctx[i:i + 1])[0] == bins[i], i
        if i % 1000 == 500:
            t0 = time.perf_counter()
            got = other.encode(bb, cc, init)[0]
            worst = max(worst, time.perf_counter() - t0)
            assert got.tobytes() == want.tobytes()
    assert h.decode_finish() == 0
    h.close()
    other.close()
    assert worst < 0.05, "a foreign engine's batch call behind a busy live kernel: %.1f ms" % (worst * 1e3)


def test_a_handle_that_outlives_its_engine_can_only_be_closed(oracle):
    """isscabac_destroy with a handle still open (and its live kernel still polling): the destroy stops the kernel and cuts the
    handle loose; every later call on it answers ISSCABAC_ERR_STATE, close frees it (once a use after free)."""
    import ctypes as C
    from isscabac_amd import Engine, Stream
    eng = Engine(0)
    init = np.zeros(3, np.uint8)
    payload = oracle.encode_stream(np.array([1, 0, 1, 1], np.uint8), np.array([0, 1, 2, 0], np.uint8), init)
    h = Stream(eng, init)
    h.decode_start(payload)
    assert h.decode_bins(np.array([0], np.uint8))[0] == 1  # leaves a live kernel polling
    L, s = eng._L, h._s
    eng.close()
    out = (C.c_uint8 * 1)()
    ctxb = (C.c_uint8 * 1)(1)
    assert L.isscabac_stream_decode_bins(s, ctxb, 1, out) == 5  # ISSCABAC_ERR_STATE
    assert L.isscabac_stream_encode_start(s) == 5
    bits = C.c_uint64(0)
    assert L.isscabac_stream_num_bits(s, C.byref(bits)) == 5
    L.isscabac_stream_close(s)
    h._s = None
