"""Host build of the engine's per-lane code (isscabac_amd/csrc/cabac_lane.h via tests/lane_harness.cpp)
against the oracle and the golden vectors.  Same inline functions the HIP kernels call, same flush /
top-up / boundary cadence, compiled with UBSan.  CPU only."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "lane_harness.cpp")
HDR = os.path.join(ROOT, "isscabac_amd", "csrc", "cabac_lane.h")
HDR2 = os.path.join(ROOT, "isscabac_amd", "csrc", "cabac_symbols.h")
OUT = os.path.join(ROOT, "tests", "_build", "liblane_harness.so")
u8p = C.POINTER(C.c_uint8)


@pytest.fixture(scope="module")
def lane():
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    if not os.path.exists(OUT) or os.path.getmtime(OUT) < max(os.path.getmtime(SRC), os.path.getmtime(HDR), os.path.getmtime(HDR2)):
        subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-shared", "-fPIC", "-fsanitize=undefined",
                               "-fno-sanitize-recover=undefined", "-I", os.path.dirname(HDR), SRC, "-o", OUT])
    L = C.CDLL(OUT)
    L.lane_encode_stream.restype = C.c_long
    L.lane_encode_stream.argtypes = [u8p, u8p, C.c_uint32, u8p, C.c_int, C.c_int, u8p, C.c_size_t, u8p,
                                     C.POINTER(C.c_int)]
    L.lane_encode_stream_pair.restype = C.c_long
    L.lane_encode_stream_pair.argtypes = [u8p, u8p, C.c_uint32, C.c_uint32, u8p, C.c_int, u8p, C.c_size_t, C.POINTER(C.c_uint32)]
    L.lane_decode_stream.restype = C.c_uint32
    L.lane_decode_stream.argtypes = [u8p, C.c_uint64, C.c_uint64, C.c_uint32, u8p, C.c_uint32, u8p, C.c_int, C.c_int,
                                     C.c_int, u8p, u8p, C.POINTER(C.c_uint32)]
    L.lane_decode_stream_out.restype = C.c_uint32
    L.lane_decode_stream_out.argtypes = [u8p, C.c_uint64, C.c_uint64, C.c_uint32, u8p, C.c_uint32, u8p, C.c_int, C.c_int,
                                         u8p, u8p, C.POINTER(C.c_uint32)]
    L.lane_max_shift_sum_ep.restype = C.c_uint32
    L.lane_max_shift_sum_ep.argtypes = [C.c_int]
    L.lane_word_shift_max.restype = C.c_uint32
    L.lane_max_shift_sum.restype = C.c_uint32
    L.lane_max_shift_sum.argtypes = [C.c_int]
    L.sym_encode_stream.restype = C.c_long
    L.sym_encode_stream.argtypes = [u8p, C.c_uint32, C.POINTER(C.c_uint32), u8p, C.c_int, u8p, C.c_size_t,
                                    C.POINTER(C.c_uint32)]
    L.sym_decode_stream.restype = C.c_uint32
    L.sym_decode_stream.argtypes = [u8p, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), u8p, C.c_int, u8p]
    L.sym_trie_check.restype = C.c_long
    L.sym_trie_check.argtypes = [u8p, C.c_uint32, C.POINTER(C.c_uint32)]
    return L


def _p(a):
    return a.ctypes.data_as(u8p)


def lane_encode(L, bins, ctx, init, flush_every):
    bins = np.ascontiguousarray(bins, np.uint8)
    ctx = np.ascontiguousarray(ctx, np.uint8)
    init = np.ascontiguousarray(init, np.uint8)
    out = np.zeros((6 * bins.size + 9 + 7) // 8 + 8, np.uint8)
    fin = np.zeros(max(1, init.size), np.uint8)
    mw = C.c_int(0)
    n = L.lane_encode_stream(_p(bins), _p(ctx), bins.size, _p(init), init.size, flush_every, _p(out), out.size,
                             _p(fin), C.byref(mw))
    assert n >= 0
    return out[:n], fin[:init.size], mw.value


def lane_decode(L, payload, ctx, init, chunk=16, group=4, pad_front=0, pad_back=0, fill=0xA5):
    payload = np.ascontiguousarray(payload, np.uint8)
    buf = np.concatenate([np.full(pad_front, fill, np.uint8), payload, np.full(pad_back, fill, np.uint8)])
    ctx = np.ascontiguousarray(ctx, np.uint8)
    init = np.ascontiguousarray(init, np.uint8)
    bins = np.zeros(max(1, ctx.size), np.uint8)
    fin = np.zeros(max(1, init.size), np.uint8)
    cnt = (C.c_uint32 * 3)()
    st = L.lane_decode_stream(_p(buf), buf.size, pad_front, payload.size, _p(ctx), ctx.size, _p(init), init.size,
                              chunk, group, _p(bins), _p(fin), cnt)
    return bins[:ctx.size], int(st), fin[:init.size], list(cnt)


def lane_decode_out(L, payload, ctx, init, lat=0, pad_front=0, pad_back=0, fill=0xA5):
    """the decode kernel's own chunk structure (lane_harness.cpp: lane_decode_stream_out)"""
    payload = np.ascontiguousarray(payload, np.uint8)
    buf = np.concatenate([np.full(pad_front, fill, np.uint8), payload, np.full(pad_back, fill, np.uint8)])
    ctx = np.ascontiguousarray(ctx, np.uint8)
    init = np.ascontiguousarray(init, np.uint8)
    bins = np.full(max(1, ctx.size), 7, np.uint8)
    fin = np.zeros(max(1, init.size), np.uint8)
    cnt = (C.c_uint32 * 3)()
    st = L.lane_decode_stream_out(_p(buf), buf.size, pad_front, payload.size, _p(ctx), ctx.size, _p(init), init.size,
                                  lat, _p(bins), _p(fin), cnt)
    return bins[:ctx.size], int(st), fin[:init.size], list(cnt)


@pytest.mark.parametrize("flush_every", [1, 2, 4, -4, 16, 17, 18, 19, 20, 21, 22])
def test_encode_golden(lane, golden, flush_every):
    for s in golden["streams"]:
        out, fin, mw = lane_encode(lane, s["bins_a"], s["ctx_a"], s["ctx_init_a"], flush_every)
        assert out.tobytes() == s["payload_a"].tobytes(), s["name"]
        assert fin.tobytes() == s["ctx_final_a"].tobytes(), s["name"]
        assert mw <= (62 if flush_every in (16, 17, 18, 19, 20, 21, 22) else 20 + 6 * min(abs(flush_every), 4) + 1)


@pytest.mark.parametrize("pads", [(0, 0, 4), (3, 0, 4), (1, 40, 4), (2, 5, 4), (7, 3, -4), (6, 2, 2), (1, 7, -2), (2, 6, 3), (3, 2, 5), (1, 4, 6)])
def test_decode_golden(lane, golden, pads):
    for s in golden["streams"]:
        bins, st, fin, _ = lane_decode(lane, s["payload_a"], s["ctx_a"], s["ctx_init_a"], pad_front=pads[0],
                                       pad_back=pads[1], group=pads[2])
        assert st == 0, s["name"]
        assert (bins == s["bins_a"]).all(), s["name"]
        assert fin.tobytes() == s["ctx_final_a"].tobytes(), s["name"]


# lat: bit 0 = the latency-oriented order; bit 1 = the payload source -- 0: PieceSrc (k_decode: aligned 16-byte pieces, every byte
# requested once), 2: WindowSrc (the pair / wide / symbol kernels: a piece at the window's own position)
@pytest.mark.parametrize("pads", [(0, 0, 0), (3, 0, 1), (1, 40, 0), (2, 5, 1), (7, 3, 0), (15, 1, 0), (16, 0, 1), (9, 33, 0),
                                  (0, 0, 2), (3, 0, 3), (1, 40, 2), (2, 5, 3), (7, 3, 2)])
def test_decode_out_golden(lane, golden, pads):
    """the kernel's chunk structure with the bin riding in bit 0 of the state byte (dec_bin_out / _nx / _mixed_nx)"""
    for s in golden["streams"]:
        bins, st, fin, _ = lane_decode_out(lane, s["payload_a"], s["ctx_a"], s["ctx_init_a"], lat=pads[2],
                                           pad_front=pads[0], pad_back=pads[1])
        assert st == 0, s["name"]
        assert (bins == s["bins_a"]).all(), s["name"]
        assert fin.tobytes() == s["ctx_final_a"].tobytes(), s["name"]


def test_decode_out_fuzz_and_corrupt(lane, oracle):
    """... fuzzed against the oracle: plain, mixed and general chunks, fixed-probability bins, corrupted payloads (bins and
    status words must be the oracle's), and no unguarded bin ever short of bits (0xDEAD.. would come back)."""
    assert [lane.lane_max_shift_sum_ep(k) for k in (1, 2, 3, 4)] == [6, 11, 17, 22]
    rng = np.random.default_rng(2024)
    paths = np.zeros(3, np.int64)
    flagged = 0
    for it in range(600):
        n = int(rng.integers(0, 2500))
        Cn = int(rng.choice([1, 3, 4, 23, 64, 100, 128, 200, 255]))
        init = rng.integers(0, 126, Cn).astype(np.uint8)
        ctx = rng.integers(0, Cn, n).astype(np.uint8)
        kind = it % 5
        if kind == 1:
            ctx[rng.random(n) < 0.5] = 0xFF
        elif kind == 2 and Cn <= 128:  # fixed-probability bins and bypass bins among the context bins
            m = rng.random(n)
            pp = rng.integers(1, 100, n)
            pp[(pp >= 46) & (pp <= 54)] = 50
            ctx = np.where(m < 0.2, 155 + pp, np.where(m < 0.3, 0xFF, ctx)).astype(np.uint8)
        elif kind == 3:  # the worst words: states 58..62, LPS runs
            init = (2 * rng.integers(58, 63, Cn) + rng.integers(0, 2, Cn)).astype(np.uint8)
        bins = rng.integers(0, 2, n).astype(np.uint8)
        if kind == 3:
            bins = np.where(rng.random(n) < 0.85, 1 - (init & 1)[np.minimum(ctx, Cn - 1)], bins).astype(np.uint8)
        elif kind != 4:
            skew = rng.random(256) ** (1 + it % 4)
            bins = (rng.random(n) < skew[ctx]).astype(np.uint8)
        want, _bits, want_fin = oracle.encode_stream(bins, ctx, init, want_final=True)
        for lat in (0, 1, 2, 3):
            dec, st, dfin, cnt = lane_decode_out(lane, want, ctx, init, lat=lat, pad_front=int(rng.integers(0, 33)),
                                                 pad_back=int(rng.integers(0, 20)))
            assert st == 0 and (dec == bins).all() and dfin.tobytes() == want_fin.tobytes(), (it, lat, hex(st))
        paths += cnt
        if n >= 8 and it % 2:
            bad = want.copy()
            mode = it % 3
            if mode == 0:
                bad[rng.integers(0, bad.size)] ^= 1 << int(rng.integers(0, 8))
            elif mode == 1:
                bad = bad[:max(1, bad.size - int(rng.integers(1, 4)))]
            else:
                bad = np.concatenate([bad, rng.integers(0, 256, 3).astype(np.uint8)])
            if bad[0] == 0xFF:
                continue  # the engine sends these to the reference-formulation decoder (RefDec)
            wdec, wst = oracle.decode_stream(bad, ctx, init)
            for src in (0, 2):
                got, st, _, _ = lane_decode_out(lane, bad, ctx, init, lat=it % 2 + src, pad_front=int(rng.integers(0, 20)), pad_back=int(rng.integers(0, 30)))
                assert st == wst and (got == wdec).all(), (it, src, hex(st), hex(wst))
            flagged += st != 0
    assert (paths > 100).all(), paths  # plain, mixed and general chunks were all exercised
    assert flagged > 50


def test_fuzz_vs_oracle(lane, oracle):
    rng = np.random.default_rng(99)
    slow = 0
    for it in range(300):
        n = int(rng.integers(0, 3000))
        Cn = int(rng.choice([1, 3, 4, 5, 23, 64, 255]))
        init = rng.integers(0, 126, Cn).astype(np.uint8)
        ctx = rng.integers(0, Cn, n).astype(np.uint8)
        if it % 4 == 1:  # (4 against the 6 modes below: every mode sees bypass bins)
            ctx[rng.random(n) < 0.5] = 0xFF
        skew = rng.random(Cn + 1) ** (1 + it % 4)
        bins = (rng.random(n) < skew[np.where(ctx == 0xFF, Cn, ctx)]).astype(np.uint8)
        want, _bits, want_fin = oracle.encode_stream(bins, ctx, init, want_final=True)
        got, fin, _ = lane_encode(lane, bins, ctx, init, (4, 16, 17, 18, 19, 20, 21, 22)[it % 8])
        assert got.tobytes() == want.tobytes()
        assert fin.tobytes() == want_fin.tobytes()
        dec, st, dfin, cnt = lane_decode(lane, want, ctx, init, pad_front=int(rng.integers(0, 9)),
                                         pad_back=int(rng.integers(0, 20)), group=(4, 16, 2, 3, 5, 6)[it % 6])
        assert st == 0 and (dec == bins).all() and dfin.tobytes() == want_fin.tobytes()
        slow += cnt[2]
    assert slow > 0  # the in-chunk refill path was exercised


def test_word_accumulator_bound(lane, oracle):
    """The plain path adds a word's four bins to `low` at once from a 32-bit accumulator (EncWordAcc): exact because four
    context bins never shift by more than 22 bits.  The bound is searched exhaustively on the product's own table, and the
    worst words (LPS runs at states 58..62) are coded through that path and compared with the oracle."""
    assert [lane.lane_max_shift_sum(k) for k in (1, 2, 3, 4)] == [6, 11, 17, 22]
    rng = np.random.default_rng(5)
    for it in range(60):
        n = 64 * int(rng.integers(1, 9))
        ctx = (np.arange(n) % 64).astype(np.uint8)
        if it % 2:
            ctx = rng.permutation(ctx)
        init = (2 * rng.integers(58, 63, 64) + rng.integers(0, 2, 64)).astype(np.uint8)  # states 58..62, any mps
        mps = init & 1
        bins = (1 - mps[ctx]).astype(np.uint8)  # the first visit of every context is an LPS
        flip = rng.random(n) < 0.1 * (it % 4)
        bins[flip] ^= 1
        want, _bits, want_fin = oracle.encode_stream(bins, ctx, init, want_final=True)
        got, fin, mw = lane_encode(lane, bins, ctx, init, 18)
        assert got.tobytes() == want.tobytes() and fin.tobytes() == want_fin.tobytes() and mw <= 62
    assert lane.lane_word_shift_max() == 22  # the worst case was among them


def test_all_lps_worst_case_width(lane, oracle, golden):
    s = {x["name"]: x for x in golden["streams"]}["all_lps_4ctx"]
    _out, _fin, mw = lane_encode(lane, s["bins_a"], s["ctx_a"], s["ctx_init_a"], 4)
    assert mw <= 44
    # the 8-bin cadence of the batch kernel at its worst: 64 contexts all at state 62, every bin an LPS (6-bit shifts)
    n = 64 * 8
    ctx = np.arange(n) % 64
    init = np.full(64, 124, np.uint8)
    bins = np.ones(n, np.uint8)  # mps is 0 -> the first LPS of every context shifts by 6
    want = oracle.encode_stream(bins, ctx, init)
    got, _fin, mw = lane_encode(lane, bins, ctx, init, 16)
    assert got.tobytes() == want.tobytes() and 50 <= mw <= 62
    # ... and the decoder's fast path at its worst: 6 bits per bin, the guard before every fourth bin fires
    for group in (2, 3, -2, 4):
        dec, st, _, _ = lane_decode(lane, want, ctx, init, pad_front=3, pad_back=1, group=group)
        assert st == 0 and (dec == bins).all()


def test_corrupt_streams_same_bins_and_status_as_oracle(lane, oracle):
    rng = np.random.default_rng(5)
    flagged = 0
    for _ in range(300):
        n = int(rng.integers(8, 800))
        Cn = int(rng.choice([1, 3, 64]))
        init = rng.integers(0, 126, Cn).astype(np.uint8)
        ctx = rng.integers(0, Cn, n).astype(np.uint8)
        if rng.random() < 0.3:
            ctx[rng.random(n) < 0.4] = 0xFF
        bins = rng.integers(0, 2, n).astype(np.uint8)
        payload = oracle.encode_stream(bins, ctx, init)
        bad = payload.copy()
        mode = rng.integers(0, 3)
        if mode == 0:
            bad[rng.integers(0, bad.size)] ^= 1 << int(rng.integers(0, 8))
        elif mode == 1:
            bad = bad[:max(1, bad.size - int(rng.integers(1, 4)))]  # truncated: the reference reads 0xFF at EOF
        else:
            bad = np.concatenate([bad, rng.integers(0, 256, 3).astype(np.uint8)])
        want, st_want = oracle.decode_stream(bad, ctx, init)
        got, st, _, _ = lane_decode(lane, bad, ctx, init, pad_front=2, pad_back=int(rng.integers(0, 30)),
                                    group=int(rng.choice([2, 3, 4, -2, 5, 6])))
        assert (got == want).all()
        assert st == st_want
        flagged += st != 0
    assert flagged > 100


def _sym_roundtrip(L, sym, params, init):
    sym = np.ascontiguousarray(sym, np.uint8)
    init = np.ascontiguousarray(init, np.uint8)
    par = (C.c_uint32 * 7)(*(tuple(params) + (0,) * (7 - len(params))))
    out = np.zeros(sym.size * 8 + 64, np.uint8)
    nb = C.c_uint32(0)
    n = L.sym_encode_stream(_p(sym), sym.size, par, _p(init), init.size, _p(out), out.size, C.byref(nb))
    assert n > 0
    dec = np.zeros(max(1, sym.size), np.uint8)
    st = L.sym_decode_stream(_p(out), n, sym.size, par, _p(init), init.size, _p(dec))
    return out[:n], nb.value, dec[:sym.size], st


def test_symbol_front_end_vs_oracle(lane, oracle):
    """binariser + context selectors + coder, host build of cabac_symbols.h, against the oracle's restatement of
    cabacDemo.m (TU, 3 contexts) and of one column of +coder/cabacEncode.m / cabacDecode.m (ISS selector)."""
    rng = np.random.default_rng(17)
    for it in range(60):
        nq = int(rng.choice([2, 3, 4, 8]))
        sym = np.minimum(np.abs(rng.normal(size=int(rng.integers(1, 500))) * nq / 2).astype(np.uint32), nq - 1)
        init = rng.integers(0, 126, 3).astype(np.uint8)
        want, want_nb = oracle.demo_encode(sym, nq, init)
        got, nb, dec, st = _sym_roundtrip(lane, sym, (0, 0, nq, 0, 0, 0), init)
        assert got.tobytes() == want.tobytes() and nb == want_nb
        assert st == 0 and (dec == sym).all()
    for it in range(120):
        nq = int(rng.choice([2, 4, 8, 16, 200]))
        method = "TU" if (it % 3 == 0 and nq <= 16) else "EG"
        k = 0 if method == "TU" else int(rng.choice([0, 0, 1, 2]))
        nlbp = int(rng.choice([1, 2, 3, 4]))
        types = int(rng.choice([15, 0, 1, 2, 4, 8, 31, 5, 26]))
        nctx = 7 * nlbp + 2
        rows = int(rng.integers(1, 420))
        sym = np.minimum(np.abs(rng.normal(size=rows) * nq / 3).astype(np.uint32), nq - 1)
        init = rng.integers(0, 126, nctx).astype(np.uint8)
        want, want_nb = oracle.iss_encode_column(sym, nq, method, k, types, nlbp, init)
        got, nb, dec, st = _sym_roundtrip(lane, sym, (0 if method == "TU" else 1, k, nq, 1, types, nlbp), init)
        assert got.tobytes() == want.tobytes(), (it, nq, method, k, nlbp, types)
        assert nb == want_nb
        assert st == 0 and (dec == sym).all()
        back, st2 = oracle.iss_decode_column(want, rows, nq, method, k, types, nlbp, init)
        assert st2 == 0 and (back == sym).all()


def test_whole_matrix_single_stream_mode(lane, oracle):
    """the reference's actual ISS mode: a whole matrix as ONE stream, columns back to back, upper neighbour
    restarting at every column (cabacEncode.m:45-70) -- host build of cabac_symbols.h with col_rows set."""
    rng = np.random.default_rng(23)
    for rows, cols, nq, method, k in [(40, 7, 8, "EG", 0), (109, 20, 8, "EG", 0), (13, 31, 4, "TU", 0), (5, 3, 16, "EG", 1)]:
        G = np.minimum(np.abs(rng.normal(size=(rows, cols)) * nq / 3).astype(np.uint32), nq - 1)
        init = rng.integers(0, 126, 23).astype(np.uint8)
        want, want_nb = oracle.iss_encode_matrix(G, nq, method, k, 15, 3, init)
        sym = G.T.ravel().astype(np.uint8)
        got, nb, dec, st = _sym_roundtrip(lane, sym, (0 if method == "TU" else 1, k, nq, 1, 15, 3, rows), init)
        assert got.tobytes() == want.tobytes() and nb == want_nb
        assert st == 0 and (dec == sym).all()
        back, st2 = oracle.iss_decode_matrix(want, rows, cols, nq, method, k, 15, 3, init)
        assert st2 == 0 and (back == G).all()
        if cols > 1:  # and it is NOT the concatenation of independently coded columns
            per_col, _ = oracle.iss_encode_column(G[:, 0], nq, method, k, 15, 3, init)
            assert got[:len(per_col)].tobytes() != per_col.tobytes() or len(got) != len(per_col)


def test_packed_helpers_ctx_quad_and_bin_collection(lane):
    """lds_ctx_quad: four state-byte offsets per ctx word; collect_bin/collected_bytes: 16 bins, two bits apiece;
    the encode kernel's 16-bit next-state table."""
    assert lane.lane_next16_mismatches() == 0
    rng = np.random.default_rng(5)
    out4 = (C.c_uint32 * 4)()
    for cw in [0, 0x7F7F7F7F, 0x03020100, 0x3F3E3D3C] + [int(x) & 0x7F7F7F7F for x in rng.integers(0, 2**32, 200)]:
        lane.lane_ctx_quad.argtypes = [C.c_uint32, C.POINTER(C.c_uint32)]
        lane.lane_ctx_quad(cw, out4)
        for k in range(4):
            c = (cw >> (8 * k)) & 0xFF
            assert out4[k] == ((c >> 2) << 8) | (c & 3)
    lane.lane_collect16.argtypes = [C.POINTER(C.c_uint32), u8p]
    for _ in range(200):
        bins = rng.integers(0, 2, 16)
        garbage = [int(g) for g in rng.integers(0, 2**32, 16, dtype=np.uint64)]
        words = (C.c_uint32 * 16)(*[(g & 0xFFFFFFFC) | (int(b) << 1) for g, b in zip(garbage, bins)])
        got = np.zeros(16, np.uint8)
        lane.lane_collect16(words, _p(got))
        assert got.tolist() == bins.tolist()


def test_reference_formulation_decoder_matches_oracle_even_on_streams_starting_with_ff(lane, oracle):
    """RefDec (used for streams whose first byte is 0xFF) against the oracle on valid, corrupted and 0xFF-led streams;
    and the look-ahead decoder differs from the oracle only on the 0xFF-led ones"""
    lane.lane_decode_stream_ref.restype = C.c_uint32
    lane.lane_decode_stream_ref.argtypes = [u8p, C.c_uint32, u8p, C.c_uint32, u8p, C.c_int, u8p]
    rng = np.random.default_rng(11)
    led, differ = 0, 0
    for it in range(400):
        n = int(rng.integers(1, 600))
        Cn = int(rng.choice([1, 3, 64]))
        init = rng.integers(0, 126, Cn).astype(np.uint8)
        ctx = rng.integers(0, Cn, n).astype(np.uint8)
        if it % 3 == 0:
            ctx[rng.random(n) < 0.3] = 0xFF
        bins = rng.integers(0, 2, n).astype(np.uint8)
        payload = oracle.encode_stream(bins, ctx, init).copy()
        if it % 2:
            payload[0] = 0xFF  # no encoder emits this
            if it % 4 == 1 and payload.size > 1:
                payload[1] = int(rng.integers(0, 256))
        elif it % 5 == 0:
            payload[rng.integers(0, payload.size)] ^= 1 << int(rng.integers(0, 8))
        want, st_want = oracle.decode_stream(payload, ctx, init)
        got = np.zeros(n, np.uint8)
        st = lane.lane_decode_stream_ref(_p(payload), payload.size, _p(ctx), n, _p(init), init.size, _p(got))
        assert (got == want).all() and st == st_want, it
        got2, st2, _, _ = lane_decode(lane, payload, ctx, init, group=3)
        if payload[0] == 0xFF:
            led += 1
            differ += int(not (got2 == want).all() or st2 != st_want)
        else:
            assert (got2 == want).all() and st2 == st_want, it
    assert led > 100 and differ > 0  # the case RefDec exists for is real


def test_symbol_code_tree_matches_the_general_walker(lane):
    """The symbol kernels' fast path walks a code tree built on the host from binarize / iss_ctx / demo_ctx (SymTrie).  For
    every parameter set the tests and the reference use: contexts and leaves along random symbol sequences equal the
    general walker's, with and without a column structure; big alphabets get no tree (general kernels)."""
    rng = np.random.default_rng(8)
    cases = []
    for nq in (2, 3, 4, 8, 13, 16):
        for method, k in ((0, 0), (1, 0), (1, 1), (1, 2)):
            for selector, types, nlbp in ((0, 0, 0), (1, 15, 3), (1, 31, 2), (1, 5, 1), (1, 0, 4), (1, 15, 8)):
                for col_rows in (0, 1, 7, 40):
                    cases.append((method, k, nq, selector, types, nlbp, col_rows))
    built = 0
    for c in cases:
        params = (C.c_uint32 * 7)(*c)
        sym = rng.integers(0, c[2], 600).astype(np.uint8)
        r = lane.sym_trie_check(_p(sym), sym.size, params)
        assert r >= 0, (c, r)
        built += r > 0
    assert built == len(cases)  # every alphabet up to 16 symbols gets a tree
    for nq in (17, 64, 200):
        params = (C.c_uint32 * 7)(1, 0, nq, 1, 15, 3, 0)
        sym = rng.integers(0, nq, 50).astype(np.uint8)
        assert lane.sym_trie_check(_p(sym), sym.size, params) == 0


def test_fixed_probability_bins_match_the_reference(lane):
    """the lane code's fixed-probability bins (enc_bin_prob / dec_bin_prob / ref_dec_bin_prob) in the batch kernels' byte
    encoding, against golden vectors produced by the reference compiled with RWTH_CABAC_FIXED_PROBABILITY"""
    import json
    from conftest import unbits, unhex
    doc = json.load(open(os.path.join(ROOT, "tests", "golden", "cabac_golden_f4.json")))
    for s in doc["streams"]:
        bins, ctx, init = unbits(s["bins"], s["n"]), unhex(s["ctx"]), unhex(s["ctx_init"])
        for mode in (1, 4, 16, 17, 18, 19, 20, 21, 22):
            got, fin, _ = lane_encode(lane, bins, ctx, init, mode)
            assert got.tobytes().hex() == s["payload"] and fin.tobytes().hex() == s["ctx_final"], (s["name"], mode)
        for group in (4, 2, 3, 5, 6):
            dec, st, dfin, _ = lane_decode(lane, unhex(s["payload"]), ctx, init, pad_front=2, pad_back=3, group=group)
            assert st == 0 and (dec == bins).all() and dfin.tobytes().hex() == s["ctx_final"], (s["name"], group)


def lane_encode_pair(L, bins, ctx, init, nmin=None):
    """the wave-pair encoder's record production + consumption (lane_harness.cpp: lane_encode_stream_pair)"""
    bins, ctx, init = (np.ascontiguousarray(a, np.uint8) for a in (bins, ctx, init))
    out = np.zeros((6 * bins.size + 9 + 7) // 8 + 64, np.uint8)
    kinds = (C.c_uint32 * 3)()
    n = L.lane_encode_stream_pair(_p(bins), _p(ctx), bins.size, bins.size if nmin is None else nmin, _p(init), init.size, _p(out),
                                  out.size, kinds)
    assert n >= 0, n
    return out[:n], list(kinds)


def test_pair_encoder_records_and_consumer_steps_equal_the_oracle(lane, oracle, golden):
    """k_encode_pair's split on the host: wave A's records (context-state pass, dummy context for bypass bins, general
    classification) and wave B's steps -- the very functions the kernel calls -- give the oracle's bytes on every golden
    stream of at most 64 contexts and on fuzzed plain, mixed, ragged and fixed-probability streams."""
    done = 0
    for s in golden["streams"]:
        if len(s["ctx_init_a"]) > 64:
            continue
        for nmin in (None, 0):  # the kernel's classification, and every chunk through the general records
            out, _ = lane_encode_pair(lane, s["bins_a"], s["ctx_a"], s["ctx_init_a"], nmin)
            assert out.tobytes() == bytes(s["payload_a"]), s["name"]
        done += 1
    assert done >= 10
    rng = np.random.default_rng(77)
    seen = [0, 0, 0]
    for it in range(300):
        nctx = int(rng.choice([1, 3, 23, 64]))
        n = int(rng.choice([0, 1, 15, 16, 17, 64, 333, 1000]))
        init = rng.integers(0, 126, nctx).astype(np.uint8)
        ctx = rng.integers(0, nctx, n).astype(np.uint8)
        mode = it % 4
        if mode >= 1:
            ctx[rng.random(n) < (0.5 if mode == 1 else 0.05)] = 0xFF
        if mode == 3 and n:  # fixed-probability bins send their chunks through the general records
            where = rng.random(n) < 0.1
            ctx[where] = (155 + rng.choice([1, 10, 30, 45, 50, 55, 70, 99], int(where.sum()))).astype(np.uint8)
        skew = rng.random(nctx + 1) ** int(rng.integers(1, 5))
        bins = (rng.random(n) < skew[np.minimum(ctx, nctx)]).astype(np.uint8)
        want = oracle.encode_stream(bins, ctx, init)
        nmin = n if it % 5 else int(rng.integers(0, n + 1))  # a shorter neighbour: the tail goes through the general records
        got, kinds = lane_encode_pair(lane, bins, ctx, init, nmin)
        assert got.tobytes() == want.tobytes(), (it, nctx, n, mode)
        seen = [a + b for a, b in zip(seen, kinds)]
    assert all(k > 50 for k in seen), seen  # plain, general and mixed chunks all ran
    assert lane.lane_word_shift_max() <= 22
