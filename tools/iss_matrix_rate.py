#!/usr/bin/env python3
"""The reference's REAL ISS format: one quantised NTF matrix = ONE bitstream, columns back to back with the contexts and the
coder carried across (ref: ISS/+coder/cabacEncode.m:45-70, cabacDecode.m:29-55) -- what matlab/+coder/cabacEncode.m and
isscabac_amd/coder.py produce, byte for byte the reference's file.  One stream is one lane: this times that serial case
(W 400 x 20 and H 109 x 20 of ISS.m:32,36, Nq = 8, EG0) next to the CPU coding the same bins.
usage: python tools/iss_matrix_rate.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401
from isscabac_amd import Engine, coder
from isscabac_amd.engine import init_by_prob

eng = Engine(0)
rng = np.random.default_rng(0)
param = {"binMethod": "DEC2EG0", "cmTypes": ("cond0", "cond1", "conds0", "conds1"), "Nlbp": 3, "equalProb": False, "fn": None}


def best(fn, n=7):
    fn()
    t = []
    for _ in range(n):
        t0 = time.perf_counter()
        out = fn()
        t.append(time.perf_counter() - t0)
    return min(t), out


for rows, cols in ((400, 20), (109, 20)):
    G = np.minimum(rng.exponential(1.2, (rows, cols)).astype(np.int64), 7).astype(np.uint8)
    te, (nbits, ci0) = best(lambda: coder.cabacEncode(G, 8, param, engine=eng))
    mp = coder._symbol_params(8, param, col_rows=rows)
    states = init_by_prob(ci0.astype(np.float64) / 255.0)
    one = np.ascontiguousarray(G.T).reshape(1, -1)
    payload, _, nb = eng.encode_symbols(one, mp, states)
    td, G2 = best(lambda: coder.cabacDecode(8, param, ci0, (rows, cols), engine=eng, payload=payload))
    assert (G2 == G).all() and 8 * payload.size == nbits
    tc, _ = best(lambda: eng.encode_symbols(one, mp, states))  # the coding call alone (cabacEncode adds the context initialisation call)
    param_eq = dict(param, equalProb=True)  # (no histogram call: the coding alone, on the host twin -- the Python mirrors' default)
    th, (hbits, hci) = best(lambda: coder.cabacEncode(G, 8, param_eq))
    from isscabac_amd import cpu_decode_symbols, cpu_encode_symbols
    thc, hout = best(lambda: cpu_encode_symbols(one, mp, states))
    assert hout[0].tobytes() == payload.tobytes()
    off = np.array([0, payload.size], np.uint64)
    thd, hsym = best(lambda: cpu_decode_symbols(payload, off, rows * cols, mp, states))
    assert (hsym[0].reshape(cols, rows).T == G).all()
    host = ("; HOST TWIN (isscabac_cpu_encode_symbols / _decode_symbols, one core, same bytes): encode %.3f ms (%.0f ns per bin), decode %.3f ms "
            "(%.0f ns per bin); coder.cabacEncode on it, equalProb %.3f ms" % (thc * 1e3, thc / int(nb[0]) * 1e9, thd * 1e3, thd / int(nb[0]) * 1e9, th * 1e3))
    line = ("%3d x %2d matrix as ONE stream: %d bins, %d bytes; cabacEncode %.2f ms (its coding call %.2f ms = %.0f ns per bin), cabacDecode %.2f ms "
            "(%.0f ns per bin)" % (rows, cols, int(nb[0]), payload.size, te * 1e3, tc * 1e3, tc / int(nb[0]) * 1e9, td * 1e3, td / int(nb[0]) * 1e9))
    try:
        from oracle import pyoracle
        orc = pyoracle.Oracle()
        t0 = time.perf_counter()
        for _ in range(20):
            want, nb_o = orc.iss_encode_matrix(G, 8, "EG", 0, 15, 3, states)
        t1 = time.perf_counter()
        assert want.tobytes() == payload.tobytes()
        line += "; one host core, C restatement of the whole walk: encode %.2f ms" % ((t1 - t0) / 20 * 1e3)
    except Exception as exc:  # noqa: BLE001
        line += "; (CPU leg skipped: %r)" % (exc,)
    print(line + host)
