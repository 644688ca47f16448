"""Batches of MORE than 255 contexts (isscabac_*_batch16: 16-bit context indices, 0xFFFF = bypass) against the reference's
own coder classes (oracle/_ref: up to 999 context models per coder, ref: CABAC/CommonDef.h:56, CABAC_ContextModelsInit.cpp:53,84)
and the C restatement: payload bytes, decoded bins and status words, on 256 / 500 / 999 contexts, ragged lengths, bypass
bins, corrupted payloads and invalid indices."""
import numpy as np
import pytest

from isscabac_amd.engine import EngineError

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    import torch  # noqa: F401
    from isscabac_amd import Engine
    return Engine(0)


def _oracle_streams(bins, ctx16, init, nb):
    """per stream: (payload bytes, decoded bins, status) by op scripts (kind 0 = context bin, 1 = bypass bin) on the compiled
    reference where it is there, else the restatement"""
    from oracle import pyoracle
    coder = pyoracle.Reference("wide") if pyoracle.have_reference() else pyoracle.Oracle()
    out = []
    for s in range(bins.shape[0]):
        n = int(nb[s])
        kind = (ctx16[s, :n] == 0xFFFF).astype(np.uint8)
        arg = np.where(kind == 1, 0, ctx16[s, :n]).astype(np.uint32)
        pay = coder.encode_ops(kind, bins[s, :n].astype(np.uint32), arg, init)[0]
        out.append(pay)
    return coder, out


@pytest.mark.parametrize("nctx", [256, 500, 999])
def test_batch16_bytes_and_bins_equal_the_reference(eng, nctx):
    rng = np.random.default_rng(nctx)
    S, B = 150, 700
    ctx16 = rng.integers(0, nctx, (S, B)).astype(np.uint16)
    ctx16[:, ::17] = nctx - 1                               # the last context is used
    ctx16[rng.random((S, B)) < 0.1] = 0xFFFF                # some bypass bins
    ctx16[:8] = np.repeat(rng.integers(0, nctx, (8, B // 10 + 1)), 10, axis=1)[:, :B]  # runs of one context
    p1 = rng.random(nctx)
    bins = (rng.random((S, B)) < p1[np.minimum(ctx16, nctx - 1)]).astype(np.uint8)
    init = rng.integers(0, 126, nctx).astype(np.uint8)      # packed states: mps | state << 1, state <= 62
    nb = rng.integers(0, B + 1, S).astype(np.uint32)
    nb[0], nb[1], nb[2] = 0, B, 1
    payload, offsets = eng.encode16(bins, ctx16, init, nb)
    coder, want = _oracle_streams(bins, ctx16, init, nb)
    for s in range(S):
        got = payload[int(offsets[s]):int(offsets[s + 1])].tobytes()
        assert got == want[s].tobytes(), "stream %d of %d contexts" % (s, nctx)
    dec, status = eng.decode16(payload, offsets, ctx16, init, nb)
    assert (status == 0).all()
    for s in range(S):
        assert (dec[s, :nb[s]] == bins[s, :nb[s]]).all() and (dec[s, nb[s]:] == 0).all()


@pytest.mark.parametrize("S,nctx,ep", [(1, 999, 0.05), (3, 256, 0.05), (65, 300, 0.01), (2, 999, 0.0), (130, 999, 0.002)])
def test_batch16_pieces_switch_between_the_straight_and_the_branching_path(eng, S, nctx, ep):
    """A piece of 16 bins in which every lane codes 16 context bins runs straight-line code, any other one per-lane branches
    (cabac_wide.hip).  Few streams with a sparse sprinkle of bypass bins make successive pieces alternate between the two --
    the bin behind a straight piece may be a bypass bin, an invalid index or the stream's end -- where a full wave with
    bypass bins never leaves the branching path and one without never enters it."""
    rng = np.random.default_rng(S * 1000 + nctx)
    B = 1000
    ctx16 = rng.integers(0, nctx, (S, B)).astype(np.uint16)
    ctx16[rng.random((S, B)) < ep] = 0xFFFF
    ctx16[:, 16::160] = 0xFFFF                               # the first bin of a piece behind a straight one
    ctx16[0, 40:44] = ctx16[0, 39]                           # a run across the word boundary
    p1 = rng.random(nctx + 1)
    bins = (rng.random((S, B)) < p1[np.minimum(ctx16, nctx)]).astype(np.uint8)
    init = rng.integers(0, 126, nctx).astype(np.uint8)
    nb = np.full(S, B, np.uint32)
    nb[-1] = 16 * 37 + 5                                     # one stream ends inside a piece
    payload, offsets = eng.encode16(bins, ctx16, init, nb)
    coder, want = _oracle_streams(bins, ctx16, init, nb)
    for s in range(S):
        assert payload[int(offsets[s]):int(offsets[s + 1])].tobytes() == want[s].tobytes(), s
    dec, status = eng.decode16(payload, offsets, ctx16, init, nb)
    assert (status == 0).all()
    for s in range(S):
        assert (dec[s, :nb[s]] == bins[s, :nb[s]]).all() and (dec[s, nb[s]:] == 0).all()


def test_batch16_corrupted_payloads_decode_like_the_reference(eng):
    from oracle import pyoracle
    rng = np.random.default_rng(5)
    nctx, S, B = 600, 96, 400
    ctx16 = rng.integers(0, nctx, (S, B)).astype(np.uint16)
    ctx16[rng.random((S, B)) < 0.2] = 0xFFFF
    bins = rng.integers(0, 2, (S, B)).astype(np.uint8)
    init = rng.integers(0, 126, nctx).astype(np.uint8)
    payload, offsets = eng.encode16(bins, ctx16, init)
    bad = payload.copy()
    for s in range(S):
        lo, hi = int(offsets[s]), int(offsets[s + 1])
        if s % 3 == 0:
            bad[lo] = 0xFF                                   # the reference's value register overflows on these
        elif s % 3 == 1:
            bad[lo + (hi - lo) // 2] ^= 0x5A
    dec, status = eng.decode16(bad, offsets, ctx16, init)
    ref = pyoracle.Reference("widec") if pyoracle.have_reference() else None
    orc = pyoracle.Oracle()
    for s in range(S):
        kind = (ctx16[s] == 0xFFFF).astype(np.uint8)
        arg = np.where(kind == 1, 0, ctx16[s]).astype(np.uint32)
        pay = bad[int(offsets[s]):int(offsets[s + 1])]
        res = orc.decode_ops(pay, kind, arg, init)  # the restatement: all three status flags
        assert (dec[s] == res[0]).all() and int(status[s]) == int(res[1]), s
        if ref is not None:  # the compiled reference: its bins and its two asserts (the over-read flag is this engine's addition)
            rres = ref.decode_ops(pay, kind, arg, init)
            assert (dec[s] == rres[0]).all() and (int(status[s]) & 3) == int(rres[1]), s
    assert (status != 0).sum() > S // 4


def test_batch16_limits_and_invalid_indices(eng):
    rng = np.random.default_rng(9)
    S, B, nctx = 70, 96, 300
    ctx16 = rng.integers(0, nctx, (S, B)).astype(np.uint16)
    bins = rng.integers(0, 2, (S, B)).astype(np.uint8)
    init = np.zeros(nctx, np.uint8)
    ref_payload, ref_offsets = eng.encode16(bins, ctx16, init)
    # an index in [nctx, 0xFFFF) leaves that stream unspecified and every other stream untouched
    c2 = ctx16.copy()
    c2[3, 10] = 40000
    p2, o2 = eng.encode16(bins, c2, init)
    for s in range(S):
        if s != 3:
            assert p2[int(o2[s]):int(o2[s + 1])].tobytes() == ref_payload[int(ref_offsets[s]):int(ref_offsets[s + 1])].tobytes()
    # <= 255 contexts give the bytes of the 8-bit call
    c8 = (ctx16 % 200).astype(np.uint16)
    p16, o16 = eng.encode16(bins, c8, np.zeros(200, np.uint8))
    p8, o8 = eng.encode(bins, c8.astype(np.uint8), np.zeros(200, np.uint8))
    assert p16.tobytes() == p8.tobytes() and (o16 == o8).all()
    with pytest.raises(EngineError, match="nctx"):
        eng.encode16(bins, ctx16, np.zeros(1000, np.uint8))
    with pytest.raises(EngineError, match="state 63"):
        eng.encode16(bins, ctx16, np.full(nctx, 127, np.uint8))
