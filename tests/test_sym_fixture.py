"""SURVEY 8 rows a14 / a15 against a SECOND derivation: tests/golden/sym_sel_fixture.json holds, for small matrices, every
symbol's bin string and every bin's context id as an independent Python restatement of the reference's MATLAB text works
them out (tests/golden/make_sym_fixture.py: no code shared with oracle/ or the kernels).  Here the oracle's C restatement
must reproduce it bin for bin -- binariser, de-binariser, the ISS and demo context selectors on the encoder side, and the
decoder side (selector on decoded bins + end-of-symbol rule) through a round trip whose bitstream is built from the
FIXTURE's (bin, context) sequence, not from the oracle's own walk.  The GPU twin is in tests/test_gpu_parity.py."""
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
METHOD = {"DEC2TU": ("TU", 0), "DEC2EG0": ("EG", 0), "DEC2EG1": ("EG", 1), "DEC2EG2": ("EG", 2)}


@pytest.fixture(scope="module")
def fx():
    with open(os.path.join(HERE, "golden", "sym_sel_fixture.json")) as f:
        return json.load(f)


def test_fixture_is_what_its_script_makes(fx, tmp_path):
    """the committed JSON is the output of the committed script (seeded, no reference needed)"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_sym_fixture", os.path.join(HERE, "golden", "make_sym_fixture.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    assert m.EGCode(0, 0) == [0] and m.EGCode(7, 0) == [1, 1, 1, 0, 0, 0, 0] and m.TUCode(3, 3) == [1, 1, 1]
    assert [len(m.EGCode(v, 0)) for v in range(8)] == [1, 3, 3, 5, 5, 5, 5, 7]
    assert len(fx["iss"]) >= 10 and len(fx["demo"]) >= 3


def test_binariser_tables(fx, oracle):
    for key, tab in fx["binarizer"].items():
        method, nq = key.split("_nq")
        m, k = METHOD[method]
        for v, want in enumerate(tab):
            got = oracle.binarize(v, m, int(nq), k)
            assert got.tolist() == want, (key, v)
            assert oracle.debinarize(np.array(want, np.uint8), m, int(nq), k) == v, (key, v)


def test_iss_selector_bin_for_bin(fx, oracle):
    for case in fx["iss"]:
        m, k = METHOD[case["method"]]
        G = np.array(case["G"])
        for c, col in enumerate(case["columns"]):
            pos, up = 0, []
            for d, nb in enumerate(col["bins_per_symbol"]):
                g = oracle.binarize(int(G[d, c]), m, case["nq"], k).tolist()
                assert g == col["bins"][pos:pos + nb], (case["name"], c, d)
                for n in range(1, nb + 1):
                    got = oracle.iss_ctx(n, g[:n - 1], up, case["types_mask"], case["nlbp"])
                    assert got - 1 == col["ctx"][pos + n - 1], (case["name"], c, d, n)  # orc_iss_ctx returns MATLAB's 1-based ctxID
                pos += nb
                up = g
            assert pos == len(col["bins"])


def test_iss_walks_code_the_fixtures_sequence(fx, oracle):
    """encoder walk: the oracle's column coder must emit exactly the stream that coding the FIXTURE's (bin, ctx) sequence
    gives; decoder walk: from that stream it must find the symbols again (its selector runs on decoded bins and its
    end-of-symbol rule decides where a symbol stops -- a wrong context or boundary anywhere derails the arithmetic decoder)"""
    rng = np.random.default_rng(7)
    for case in fx["iss"]:
        m, k = METHOD[case["method"]]
        nctx = 7 * case["nlbp"] + 2
        init = rng.integers(0, 126, nctx).astype(np.uint8)
        G = np.array(case["G"])
        for c, col in enumerate(case["columns"]):
            bins, ctx = np.array(col["bins"], np.uint8), np.array(col["ctx"], np.uint8)
            want = oracle.encode_stream(bins, ctx, init)
            got, nb = oracle.iss_encode_column(G[:, c], case["nq"], m, k, case["types_mask"], case["nlbp"], init)
            assert nb == bins.size and got.tobytes() == want.tobytes(), (case["name"], c)
            sym, st = oracle.iss_decode_column(want, case["rows"], case["nq"], m, k, case["types_mask"], case["nlbp"], init)
            assert st == 0 and sym.tolist() == G[:, c].tolist(), (case["name"], c)
        # the whole matrix as one stream (cabacEncode.m:45-70): the columns' sequences back to back
        bins = np.array([b for col in case["columns"] for b in col["bins"]], np.uint8)
        ctx = np.array([x for col in case["columns"] for x in col["ctx"]], np.uint8)
        want = oracle.encode_stream(bins, ctx, init)
        got, nb = oracle.iss_encode_matrix(G, case["nq"], m, k, case["types_mask"], case["nlbp"], init)
        assert nb == bins.size and got.tobytes() == want.tobytes(), case["name"]
        Gd, st = oracle.iss_decode_matrix(want, case["rows"], case["cols"], case["nq"], m, k, case["types_mask"], case["nlbp"], init)
        assert st == 0 and (Gd == G).all(), case["name"]


def test_demo_selector(fx, oracle):
    rng = np.random.default_rng(8)
    for case in fx["demo"]:
        init = rng.integers(0, 126, 3).astype(np.uint8)
        bins, ctx = np.array(case["bins"], np.uint8), np.array(case["ctx"], np.uint8)
        want = oracle.encode_stream(bins, ctx, init)
        got, nb = oracle.demo_encode(np.array(case["symbols"]), case["nq"], init)
        assert nb == bins.size and got.tobytes() == want.tobytes(), case["name"]
        sym, st = oracle.demo_decode(want, len(case["symbols"]), case["nq"], init)
        assert st == 0 and sym.tolist() == case["symbols"], case["name"]
