"""The MEX gateway (matlab/SimpleCABACMex.cpp) compiled against a stub mex.h: the reference's command strings,
arities, return shapes and error texts (ref: CABAC/SimpleCABACMex.cpp:100-472), plus the batch commands."""
import numpy as np
import pytest

from mex_gateway import Gateway, MexError


@pytest.fixture(scope="module")
def gw():
    return Gateway()


def test_compiles_and_rejects_like_the_reference(gw):
    with pytest.raises(MexError, match="input 0 must be a valid keyword"):
        gw.call()
    with pytest.raises(MexError, match="input 0 must be a valid keyword"):
        gw.call(np.array([1.0]))
    with pytest.raises(MexError, match="Invalid Command"):
        gw.call("noSuchCommand")
    with pytest.raises(MexError, match="Invalid Command"):
        gw.call("getEncoderStats", np.array([1.0]), 0.0)  # Windows-only in the reference; Linux falls through
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
def test_per_bin_commands_round_trip_and_numbits(gw, oracle, golden, tmp_path):
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
