"""The C restatement (oracle/cabac_oracle.c) against the golden vectors produced by the real reference.

Runs anywhere (no GPU, no /root/reference): this is what pins the oracle on the GPU box too.
"""
import hashlib

import numpy as np

from conftest import unhex, unbits


def test_streams_bytes_bits_and_states(oracle, golden):
    for s in golden["streams"]:
        payload, bits, fin = oracle.encode_stream(s["bins_a"], s["ctx_a"], s["ctx_init_a"], want_final=True)
        assert payload.tobytes() == s["payload_a"].tobytes(), s["name"]
        assert bits == s["bits_before_finish"], s["name"]
        assert fin.tobytes() == s["ctx_final_a"].tobytes(), s["name"]


def test_streams_decode(oracle, golden):
    for s in golden["streams"]:
        bins, status = oracle.decode_stream(s["payload_a"], s["ctx_a"], s["ctx_init_a"])
        assert status == 0, s["name"]
        assert (bins == s["bins_a"]).all(), s["name"]


def test_size_identity(golden):
    # SURVEY.md hard parts: file_bytes == ceil((sum of renorm shifts + 9) / 8); here only the
    # weaker, input-independent bound the engine sizes its slots with: ceil((6*ctx + ep + 9)/8)
    for s in golden["streams"]:
        n_ep = int((s["ctx_a"] == 0xFF).sum())
        n_ctx = s["n"] - n_ep
        assert len(s["payload_a"]) <= (6 * n_ctx + n_ep + 9 + 7) // 8, s["name"]


def test_op_scripts(oracle, golden):
    for o in golden["ops"]:
        init = unhex(o["ctx_init"])
        payload, fin = oracle.encode_ops(o["kind"], o["val"], o["arg"], init)
        assert payload.tobytes().hex() == o["payload"], o["name"]
        assert fin.tobytes().hex() == o["ctx_final"], o["name"]
        res, status, dfin = oracle.decode_ops(unhex(o["payload"]), o["kind"], o["arg"], init)
        assert status == 0 and list(res) == o["val"], o["name"]
        assert dfin.tobytes().hex() == o["ctx_final"], o["name"]


def test_known_answers_quoted_in_survey(golden):
    by = {s["name"]: s for s in golden["streams"]}
    assert by["empty"]["payload"] == "fe80"
    assert by["lcg200_3ctx"]["payload"] == "32fd58c04f5b8d5191c90bc45167388163df0fa895e266e49e9590"
    assert by["lcg200_3ctx"]["bits_before_finish"] == 184
    assert golden["ops"][0]["payload"] == "abb12f09"


def test_state_tables(oracle, golden):
    for p in range(128):
        assert oracle.lib.orc_ctx_next_mps(p) == golden["next_state_mps"][p]
        assert oracle.lib.orc_ctx_next_lps(p) == golden["next_state_lps"][p]
    # renorm == clz32(lps) - 23 for every reachable LPS value (states 0..62)
    for state in range(63):
        for q in range(4):
            lps = oracle.lib.orc_lps_table(state, q)
            assert 6 <= lps <= 240
            assert oracle.lib.orc_renorm_table(lps >> 3) == (32 - int(lps).bit_length()) - 23
            assert 256 <= (lps << oracle.lib.orc_renorm_table(lps >> 3)) < 512


def test_synthetic_generator_and_coder(oracle, golden):
    for g in golden["synth"]:
        bins, ctx = oracle.synth(g["seed"], g["s0"], g["nS"], g["B"], g["C"], bool(g["ep_half"]))
        assert hashlib.sha256(bins.tobytes()).hexdigest() == g["sha256_bins"]
        assert hashlib.sha256(ctx.tobytes()).hexdigest() == g["sha256_ctx"]
        assert (bins[0][:64] == unbits(g["first_bins"], 64)).all()
        assert ctx[0][:64].tobytes().hex() == g["first_ctx"]
        init = np.zeros(g["C"], np.uint8)
        for s in range(g["nS"]):
            assert oracle.encode_stream(bins[s], ctx[s], init).tobytes().hex() == g["payloads"][s]


def test_p0_map_anchors(oracle):
    # CABAC_ContextModelsInit.cpp:124-148 cannot be compiled here (mex.h); anchors are the values the
    # survey's probe printed from the reference plus structural properties.
    def ms(k):
        p = oracle.p0_to_packed(k / 255.0)
        return (p & 1, p >> 1)
    for k in range(0, 5):
        assert ms(k) == (1, 62)
    for k in (125, 126, 127):
        assert ms(k) == (1, 0)
    for k in (128, 129, 130):
        assert ms(k) == (0, 0)
    for k in range(131, 138):
        assert ms(k) == (0, 1)
    for k in range(251, 256):
        assert ms(k) == (0, 62)
    states = [ms(k)[1] for k in range(128, 256)]
    assert states == sorted(states)
    for k in range(256):  # symmetric in p0 <-> 1-p0 up to the p0 == 0.5 tie
        a, b = ms(k), ms(255 - k)
        assert a[1] == b[1] and a[0] != b[0]
    assert oracle.init_by_prob([0.5, 0.5, 0.5]).tolist() == [0, 0, 0]  # cabacDemo.m:49
    assert oracle.init_by_state([7, 1, 20, 9, 0, 3]).tolist() == [41, 6]  # ctxIdx slot ignored


def test_batch_helpers(oracle):
    bins, ctx = oracle.synth(0xCABAC003, 5, 6, 300, 64)
    init = np.zeros(64, np.uint8)
    nb = np.array([300, 0, 1, 299, 17, 300], np.uint32)
    payload, offsets = oracle.encode_batch(bins, ctx, init, nb)
    for s in range(6):
        ref = oracle.encode_stream(bins[s][:nb[s]], ctx[s][:nb[s]], init)
        assert payload[int(offsets[s]):int(offsets[s + 1])].tobytes() == ref.tobytes()
    dec, status = oracle.decode_batch(payload, offsets, ctx, init, nb)
    assert (status == 0).all()
    for s in range(6):
        assert (dec[s][:nb[s]] == bins[s][:nb[s]]).all()


def test_multi_ep_equals_single_ep(oracle, golden):
    """encodeBinsEP(pattern, n) (ref: CABAC_ArithmeticEncoder.cpp:278-319) writes the same bytes as n calls of
    encodeBinEP: the batch ABI therefore needs only the per-bin bypass marker.  Checked on the golden op scripts,
    whose expected payloads come from the reference running the multi-bin form."""
    for o in golden["ops"]:
        bins, ctx = [], []
        for k, v, a in zip(o["kind"], o["val"], o["arg"]):
            if k == 0:
                bins.append(v)
                ctx.append(a)
            elif k == 1:
                bins.append(v)
                ctx.append(0xFF)
            else:
                for i in range(a - 1, -1, -1):
                    bins.append((v >> i) & 1)
                    ctx.append(0xFF)
        payload = oracle.encode_stream(bins, ctx, unhex(o["ctx_init"]))
        assert payload.tobytes().hex() == o["payload"], o["name"]
        dec, status = oracle.decode_stream(payload, ctx, unhex(o["ctx_init"]))
        assert status == 0 and dec.tolist() == bins


def test_p0_map_engine_oracle_and_independent_fixture_agree(oracle):
    """row a9: the engine's host-side isscabac_init_by_prob, the oracle's restatement and a 256-entry table derived
    independently (numpy, double: tests/golden/make_p0_fixture.py) give the same (mps, state) for every p0 = k / 255"""
    import ctypes
    import json
    import os
    from isscabac_amd import build
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    want = np.array(json.load(open(os.path.join(root, "tests", "golden", "p0_state_lut.json")))["packed"], np.uint8)
    p0 = np.arange(256, dtype=np.float64) / 255.0
    assert (oracle.init_by_prob(p0) == want).all()
    lib = ctypes.CDLL(build.build_lib())  # a pure host function: no GPU involved
    got = np.zeros(256, np.uint8)
    rc = lib.isscabac_init_by_prob(p0.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), 256,
                                   got.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)))
    assert rc == 0 and (got == want).all()
    # the shape SURVEY 8(a)-9 quotes from its probe of the real code
    assert want[0] == (62 << 1 | 1) and want[127] == (0 << 1 | 1) and want[128] == 0 and want[255] == 62 << 1


def test_iss_context_init_against_the_independent_restatement(oracle):
    """cabacInitContextModel.m is MATLAB (no run of the reference possible here), so the oracle's C
    restatement is pinned by a second one written apart from it -- Python lists for the cell arrays, straight from the .m
    text (tests/golden/make_f1_fixture.py): every probability must agree to the last bit."""
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    fx = json.load(open(os.path.join(root, "tests", "golden", "iss_ctx_init_fixture.json")))
    assert len(fx["cases"]) >= 8
    for c in fx["cases"]:
        G = np.array(c["G"])
        p0, _b = oracle.iss_init_context_model(G, c["nq"], c["method"], c["k"], c["types"], c["nlbp"])
        assert p0.tolist() == c["p0"], (G.shape, c["method"], c["types"])
