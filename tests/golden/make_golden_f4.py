#!/usr/bin/env python3
"""Generate tests/golden/cabac_golden_f4.json from the REAL reference built with its two compile-time switches on
(oracle/_ref/libcabac_ref_full.so: RWTH_CABAC_FIXED_PROBABILITY and RWTH_TRACE_CABAC_STATES, see oracle/Makefile).

Run in the build container (needs /root/reference):

    python tests/golden/make_golden_f4.py

Every expected payload, decoded value, trace row and transition count comes from the reference classes
(CABAC_ArithmeticEncoder::encodeBinProb / CABAC_ArithmeticDecoder::decodeBinProb, ContextModel::addCabacStep) driven by
oracle/ref_shim.cpp.  The fixture is data only: inputs and expected outputs.

Fixed-probability values 46..49 and 51..54 are left out on purpose: with those rows of sm_aucLPSTProbTable the
reference's own MPS path can leave m_uiRange below 256 (LPS > range / 2, CABAC_ArithmeticEncoder.cpp:208-232), after
which its decoder indexes sm_aucLPSTable out of bounds on the next context bin (CABAC_ArithmeticDecoder.cpp:94).
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.pyoracle import PROB_BASE, Reference  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cabac_golden_f4.json")
SAFE_PROBS = [p for p in range(1, 100) if not (46 <= p <= 49 or 51 <= p <= 54)]


def hx(a):
    return np.ascontiguousarray(a, np.uint8).tobytes().hex()


def bits_hex(b):
    return np.packbits(np.asarray(b, np.uint8)).tobytes().hex()


def sparse(trans):
    ii, jj = np.nonzero(trans)
    return [[int(i), int(j), int(trans[i, j])] for i, j in zip(ii, jj)]


def main():
    ref = Reference("gold4", full=True)
    rng = np.random.default_rng(20261004)
    ops, streams, traces = [], [], []

    # ---- the reference's own test program with its fixed-probability part, CABAC/SimpleCABAC.cpp:63-103,136-172 ------
    kind = [0] * 6 + [1] * 5 + [2] + [0] * 6 + [3] * 6
    val = [0, 0, 1, 0, 1, 1, 1, 0, 0, 1, 0, 18, 1, 1, 0, 1, 1, 1, 1, 0, 0, 1, 1, 0]
    arg = [0] * 6 + [0] * 5 + [5] + [1] * 6 + [10, 10, 10, 30, 30, 30]
    init = np.array([(20 << 1) | 0, (0 << 1) | 1], np.uint8)
    payload, fin = ref.encode_ops(kind, val, arg, init)
    res, st, _ = ref.decode_ops(payload, kind, arg, init)
    assert st == 0 and list(res) == val
    ops.append(dict(name="simplecabac_main_fixed_prob", note="CABAC/SimpleCABAC.cpp:48-116 with RWTH_CABAC_FIXED_PROBABILITY",
                    kind=kind, val=val, arg=arg, ctx_init=hx(init), payload=hx(payload), ctx_final=hx(fin)))

    # ---- every usable probability once, both bin values ----------------------------------------------------------------
    for bin_ in (0, 1):
        k = [3] * len(SAFE_PROBS)
        v = [bin_] * len(SAFE_PROBS)
        a = list(SAFE_PROBS)
        payload, fin = ref.encode_ops(k, v, a, np.zeros(1, np.uint8))
        res, st, _ = ref.decode_ops(payload, k, a, np.zeros(1, np.uint8))
        assert st == 0 and list(res) == v
        ops.append(dict(name="every_prob_bin%d" % bin_, note="one bin per probability 1..45, 50, 55..99", kind=k, val=v, arg=a,
                        ctx_init=hx(np.zeros(1, np.uint8)), payload=hx(payload), ctx_final=hx(fin)))

    # ---- random scripts: context bins, EP bins, multi-EP, fixed-probability bins ---------------------------------------
    for t in range(8):
        k, v, a = [], [], []
        for _ in range(int(rng.integers(40, 400))):
            r = rng.integers(0, 4)
            if r == 0:
                k.append(0); v.append(int(rng.integers(0, 2))); a.append(int(rng.integers(0, 5)))
            elif r == 1:
                k.append(1); v.append(int(rng.integers(0, 2))); a.append(0)
            elif r == 2:
                nb = int(rng.integers(1, 33))
                k.append(2); v.append(int(rng.integers(0, 2 ** nb))); a.append(nb)
            else:
                p = int(rng.choice(SAFE_PROBS))
                k.append(3); v.append(int(rng.random() * 100 < p)); a.append(p)
        init5 = rng.integers(0, 126, 5).astype(np.uint8)
        payload, fin = ref.encode_ops(k, v, a, init5)
        res, st, dfin = ref.decode_ops(payload, k, a, init5)
        assert st == 0 and list(res) == v and (fin == dfin).all()
        ops.append(dict(name="mixed_prob_ops_%d" % t, note="ctx / EP / multi-EP / fixed-probability mix", kind=k, val=v, arg=a,
                        ctx_init=hx(init5), payload=hx(payload), ctx_final=hx(fin)))

    # ---- streams in the engine's byte encoding: ctx byte 155 + p = fixed-probability bin, 0xFF = EP bin ----------------
    for t, (n, C, share) in enumerate([(1, 1, 1.0), (16, 3, 0.5), (333, 8, 0.3), (2048, 64, 0.25), (4096, 64, 0.6),
                                       (1500, 128, 0.1)]):
        init_c = rng.integers(0, 126, C).astype(np.uint8)
        ctx = rng.integers(0, C, n).astype(np.uint8)
        bins = rng.integers(0, 2, n).astype(np.uint8)
        isp = rng.random(n) < share
        probs = rng.choice(SAFE_PROBS, n)
        ctx[isp] = (PROB_BASE + probs[isp]).astype(np.uint8)
        bins[isp] = (rng.random(int(isp.sum())) * 100 < probs[isp]).astype(np.uint8)
        ep = (~isp) & (rng.random(n) < 0.2)
        ctx[ep] = 0xFF
        kind = np.where(ctx == 0xFF, 1, np.where(ctx > PROB_BASE, 3, 0)).astype(np.uint8)
        arg = np.where(kind == 3, ctx.astype(np.int64) - PROB_BASE, np.where(kind == 1, 0, ctx)).astype(np.uint32)
        payload, fin = ref.encode_ops(kind, bins.astype(np.uint32), arg, init_c)
        res, st, dfin = ref.decode_ops(payload, kind, arg, init_c)
        assert st == 0 and (res == bins).all() and (fin == dfin).all()
        streams.append(dict(name="prob_stream_%d_n%d_c%d" % (t, n, C), n=n, nctx=C, ctx_init=hx(init_c), bins=bits_hex(bins),
                            ctx=hx(ctx), payload=hx(payload), ctx_final=hx(fin)))

    # ---- state traces: getEncoderStats / getDecoderStats per context -----------------------------------------------------
    for t, (n, C) in enumerate([(1, 1), (40, 3), (600, 3), (3000, 8), (5000, 23)]):
        init_c = rng.integers(0, 126, C).astype(np.uint8)
        ctx = rng.integers(0, C, n).astype(np.uint8)
        p1 = rng.random(C) ** 2
        bins = (rng.random(n) < p1[ctx]).astype(np.uint8)
        ctx[rng.random(n) < 0.1] = 0xFF  # EP bins leave no trace
        kind = (ctx == 0xFF).astype(np.uint8)
        arg = np.where(kind == 1, 0, ctx).astype(np.uint32)
        payload, fin = ref.encode_ops(kind, bins.astype(np.uint32), arg, init_c)
        enc = [ref.get_trace(0, c) for c in range(C)]
        res, st, _ = ref.decode_ops(payload, kind, arg, init_c)
        assert st == 0 and (res == bins).all()
        dec = [ref.get_trace(1, c) for c in range(C)]
        traces.append(dict(name="trace_%d_n%d_c%d" % (t, n, C), n=n, nctx=C, ctx_init=hx(init_c), bins=bits_hex(bins), ctx=hx(ctx),
                           payload=hx(payload),
                           encoder=[dict(trace=hx(tr), transitions=sparse(tt)) for tr, tt in enc],
                           decoder=[dict(trace=hx(tr), transitions=sparse(tt)) for tr, tt in dec]))

    doc = dict(meta=dict(source="oracle/_ref/libcabac_ref_full.so = /root/reference/CABAC/*.cpp compiled with -D_WIN32 "
                                "(RWTH_TRACE_CABAC_STATES) and RWTH_CABAC_FIXED_PROBABILITY flipped to 1 (oracle/Makefile)",
                         generator="tests/golden/make_golden_f4.py", prob_base=PROB_BASE,
                         trace_row="[codedBin, trace state before, mps before, trace state after, mps after]; the decoder "
                                   "side's codedBin is always 0 (SimpleCABACMex.cpp:316-327 stores it before decoding)"),
               ops=ops, streams=streams, traces=traces)
    with open(OUT, "w") as f:
        json.dump(doc, f, separators=(",", ":"))
    print("wrote %s: %d op scripts, %d streams, %d trace cases, %d bytes" % (OUT, len(ops), len(streams), len(traces),
                                                                             os.path.getsize(OUT)))


if __name__ == "__main__":
    main()
