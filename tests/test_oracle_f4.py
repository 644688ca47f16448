"""SURVEY 8(f)-4 on the CPU: the oracle's fixed-probability bins (encodeBinProb / decodeBinProb), state traces
(getEncoderStats / getDecoderStats) and TRk / FL32 binarisers.

Pinning: tests/golden/cabac_golden_f4.json was produced by the reference's own classes compiled with its two compile-time
switches on (oracle/_ref/libcabac_ref_full.so, recipe in oracle/Makefile); where that build is present (the build
container) the restatement is additionally fuzzed against it directly.  TRk / FL32 are restated from the MATLAB text
(CABAC/cabacBinarizer.m:39-54,71-75; no MATLAB in the image): text-pinned, checked here against hand-derived strings."""
import json
import os

import numpy as np
import pytest

from conftest import unbits, unhex

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAFE_PROBS = [p for p in range(1, 100) if not (46 <= p <= 49 or 51 <= p <= 54)]


@pytest.fixture(scope="module")
def golden4():
    with open(os.path.join(ROOT, "tests", "golden", "cabac_golden_f4.json")) as f:
        return json.load(f)


@pytest.fixture(scope="module")
def reference_full():
    from oracle import pyoracle
    if not (pyoracle.have_reference_full() or os.path.isdir("/root/reference/CABAC")):
        pytest.skip("oracle/_ref/libcabac_ref_full.so not available here")
    return pyoracle.Reference("f4", full=True)


def dense(sp):
    t = np.zeros((128, 128), np.uint32)
    for i, j, n in sp:
        t[i, j] = n
    return t


def test_prob_op_scripts_match_the_reference(oracle, golden4):
    assert golden4["ops"][0]["payload"] == "abb12f03df"  # the reference's own SimpleCABAC.cpp run with the flag on
    for op in golden4["ops"]:
        init = unhex(op["ctx_init"])
        payload, fin = oracle.encode_ops(op["kind"], op["val"], op["arg"], init)
        assert payload.tobytes().hex() == op["payload"], op["name"]
        assert fin.tobytes().hex() == op["ctx_final"], op["name"]
        res, st, dfin = oracle.decode_ops(unhex(op["payload"]), op["kind"], op["arg"], init)
        assert st == 0 and res.tolist() == op["val"] and dfin.tobytes().hex() == op["ctx_final"], op["name"]


def test_prob_streams_in_the_engines_byte_encoding(oracle, golden4):
    for s in golden4["streams"]:
        bins, ctx, init = unbits(s["bins"], s["n"]), unhex(s["ctx"]), unhex(s["ctx_init"])
        payload, _bits, fin = oracle.encode_stream(bins, ctx, init, want_final=True)
        assert payload.tobytes().hex() == s["payload"], s["name"]
        assert fin.tobytes().hex() == s["ctx_final"], s["name"]
        dec, st = oracle.decode_stream(unhex(s["payload"]), ctx, init)
        assert st == 0 and (dec == bins).all(), s["name"]


def test_state_traces_match_the_reference(oracle, golden4):
    for t in golden4["traces"]:
        bins, ctx, init = unbits(t["bins"], t["n"]), unhex(t["ctx"]), unhex(t["ctx_init"])
        for side, key in ((False, "encoder"), (True, "decoder")):
            for c in range(t["nctx"]):
                trace, trans = oracle.trace_stats(bins, ctx, init, c, decoder=side)
                assert trace.tobytes().hex() == t[key][c]["trace"], (t["name"], key, c)
                assert (trans == dense(t[key][c]["transitions"])).all(), (t["name"], key, c)
                assert int(trans.sum()) == trace.shape[0] == int(((ctx == c)).sum())
                if side:
                    assert not trace[:, 0].any()  # the gateway's decoder-side quirk
                else:
                    assert (trace[:, 0] == bins[ctx == c]).all()


def test_trace_state_remap(oracle):
    # ContextModel.cpp:128-129: mps 0 -> 63 - state (63 .. 1), mps 1 -> state + 64 (64 .. 126)
    for state in range(63):
        assert oracle.lib.orc_trace_state((state << 1) | 0) == 63 - state
        assert oracle.lib.orc_trace_state((state << 1) | 1) == 64 + state


def test_prob_table_shape(oracle):
    # sm_aucLPSTProbTable: 49 rows, decreasing down the rows, increasing along the columns; first and last rows as printed
    t = np.array([[oracle.lib.orc_prob_table(r, q) for q in range(4)] for r in range(49)])
    assert t[0].tolist() == [139, 171, 202, 234] and t[48].tolist() == [4, 4, 4, 5]
    assert (np.diff(t, axis=0) <= 0).all() and (np.diff(t, axis=1) >= 0).all()
    # rows 0..3 are the ones whose column-0 LPS exceeds half of the smallest range (256): the probabilities left out
    assert (t[:4, 0] > 128).all() and (t[4:, 0] <= 128).all()


def test_prob_fuzz_against_the_compiled_reference(oracle, reference_full):
    rng = np.random.default_rng(4)
    for it in range(200):
        k, v, a = [], [], []
        for _ in range(int(rng.integers(1, 300))):
            r = rng.integers(0, 3)
            if r == 0:
                k.append(0); v.append(int(rng.integers(0, 2))); a.append(int(rng.integers(0, 4)))
            elif r == 1:
                k.append(1); v.append(int(rng.integers(0, 2))); a.append(0)
            else:
                k.append(3); v.append(int(rng.integers(0, 2))); a.append(int(rng.choice(SAFE_PROBS)))
        init = rng.integers(0, 126, 4).astype(np.uint8)
        want, wfin = reference_full.encode_ops(k, v, a, init)
        got, gfin = oracle.encode_ops(k, v, a, init)
        assert got.tobytes() == want.tobytes() and (gfin == wfin).all()
        res, st, _ = oracle.decode_ops(want, k, a, init)
        assert st == 0 and res.tolist() == v


def test_trace_fuzz_against_the_compiled_reference(oracle, reference_full):
    rng = np.random.default_rng(5)
    for it in range(30):
        n, C = int(rng.integers(0, 2000)), int(rng.choice([1, 2, 3, 23]))
        init = rng.integers(0, 126, C).astype(np.uint8)
        ctx = rng.integers(0, C, n).astype(np.uint8)
        bins = (rng.random(n) < rng.random(C)[ctx] ** 3).astype(np.uint8)
        kind = np.zeros(n, np.uint8)
        payload, _ = reference_full.encode_ops(kind, bins.astype(np.uint32), ctx.astype(np.uint32), init)
        for c in range(C):
            want_t, want_x = reference_full.get_trace(0, c)
            got_t, got_x = oracle.trace_stats(bins, ctx, init, c)
            assert got_t.tobytes() == want_t.tobytes() and (got_x == want_x).all()
        reference_full.decode_ops(payload, kind, ctx.astype(np.uint32), init)
        for c in range(C):
            want_t, want_x = reference_full.get_trace(1, c)
            got_t, got_x = oracle.trace_stats(bins, ctx, init, c, decoder=True)
            assert got_t.tobytes() == want_t.tobytes() and (got_x == want_x).all()


def test_tr_and_fl_binarisers_text_pinned(oracle):
    """cabacBinarizer.m:39-54 (TRCode) and :71-75 (FLCode), derived by hand from the MATLAB text."""
    b = lambda v, m, nq=8, k=0: oracle.binarize(v, m, nq=nq, k=k).tolist()
    # TR1, maxVal 7: n_p = floor(v / 2) + 1 ones-then-zero prefix, one suffix bit
    assert b(0, "TR", k=1) == [0, 0]
    assert b(1, "TR", k=1) == [0, 1]
    assert b(5, "TR", k=1) == [1, 1, 0, 1]
    assert b(6, "TR", k=1) == [1, 1, 1, 0, 0]
    assert b(7, "TR", k=1) == [1, 1, 1, 0, 1]  # v >= maxVal: the suffix stays all ones (the TODO branch)
    # TR0 is unary with a terminating zero for every value, the maximum included (unlike TU)
    assert b(3, "TR", nq=4, k=0) == [1, 1, 1, 0] and b(0, "TR", nq=4, k=0) == [0]
    # TR2
    assert b(6, "TR", nq=16, k=2) == [1, 0, 1, 0] and b(15, "TR", nq=16, k=2) == [1, 1, 1, 0, 1, 1]
    assert b(11, "TR", nq=12, k=2) == [1, 1, 0, 1, 1]  # the maximum: all-ones suffix happens to equal its remainder
    assert b(12, "TR", nq=12, k=2) == [1, 1, 1, 0, 1, 1]  # above the maximum: ones, not the remainder 0
    # FL32: 32 bins, MSB first
    assert b(0, "FL32") == [0] * 32
    assert b(5, "FL32") == [0] * 29 + [1, 0, 1]
    assert b(255, "FL32") == [0] * 24 + [1] * 8
    # TU / EG unchanged
    assert b(7, "TU") == [1] * 7 and b(3, "EG", k=0) == [1, 1, 0, 0, 0]
