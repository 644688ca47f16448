#!/usr/bin/env python3
"""cabacInitContextModel.m (ISS/+coder, lines 1-130) restated INDEPENDENTLY of this repository's C code -- plain Python
lists standing in for MATLAB's cell arrays, statement by statement from the reference's text -- and run on a few small
matrices to produce tests/golden/iss_ctx_init_fixture.json.  The reference function is MATLAB (no MATLAB in this image),
so this second derivation, not a run of the reference, is what pins row f1 of SURVEY 8 beside oracle/cabac_oracle.c's
own restatement: the two were written apart and must agree to the last bit (tests/test_oracle_golden.py).

    binarisation: CABAC/cabacBinarizer.m:30-69 (TUCode, EGCode, FLCode)
    statistics:   ISS/+coder/cabacInitContextModel.m:14-129; types as in ISS/ISS.m:51 plus 'condbinlft'
"""
import json
import math
import os

import numpy as np

NAN = "nan"  # the cells of the shifted-in first row: a scalar NaN (length 1, compares false with everything)


def fl_code(v, nbits):  # cabacBinarizer.m:71-75
    return [(v // (1 << (nbits - i))) % 2 for i in range(1, nbits + 1)]


def tu_code(v, max_val):  # :30-37
    if v == max_val:
        return [1] * max_val
    return [1] * v + [0]


def eg_code(v, k):  # :56-69
    n_p = int(math.floor(math.log2(v / 2 ** k + 1))) + 1
    n_s = k + n_p - 1
    c = [1] * (n_p + n_s)
    c[n_p - 1] = 0
    v_s = v - 2 ** k * (2 ** (n_p - 1) - 1)
    c[n_p:] = fl_code(v_s, n_s)
    return c


def at(x, i):
    """x(i), 1-based; a NaN cell yields NaN (None here)"""
    if x == NAN:
        assert i == 1
        return None
    return x[i - 1]


def mean_or(mask, default):
    return sum(mask) / len(mask) if len(mask) else default


def norm_of(mask):  # "if ~isempty(mask) && sum(mask) ~= 0, norm = sum(mask) / length(mask); else, norm = 1; end"
    return sum(mask) / len(mask) if len(mask) and sum(mask) != 0 else 1.0


def init_context_model(G, binarize, types, nlbp):
    D, K = len(G), len(G[0])
    gbin = [[binarize(G[d][k]) for k in range(K)] for d in range(D)]
    gbin_up1 = [[NAN] * K] + gbin[:-1]
    def first_zero_or_len(x):
        return x.index(0) + 1 if 0 in x else len(x)
    np_ = [[first_zero_or_len(x) for x in row] for row in gbin]
    np_up1 = [[0] * K] + np_[:-1]
    L = [[len(x) for x in row] for row in gbin]
    L_up1 = [[1 if x == NAN else len(x) for x in row] for row in gbin_up1]
    cells = [(d, k) for k in range(K) for d in range(D)]  # MATLAB's linear (column-major) order; means do not depend on it
    p_pre, pc0, pc1, p_lft, p_suf, pcs0, pcs1 = ([0.0] * nlbp for _ in range(7))
    for n in range(1, nlbp + 1):
        sel = [c for c in cells if n <= np_[c[0]][c[1]]]
        p_pre[n - 1] = mean_or([at(gbin[d][k], n) == 0 for d, k in sel], 0.0)
        sel = [c for c in cells if n <= np_[c[0]][c[1]] and n <= np_up1[c[0]][c[1]]]
        if "cond0" in types:
            joint = mean_or([at(gbin[d][k], n) == 0 and at(gbin_up1[d][k], n) == 0 for d, k in sel], 0.0)
            pc0[n - 1] = joint / norm_of([at(gbin_up1[d][k], n) == 0 for d, k in sel])
        if "cond1" in types:
            joint = mean_or([at(gbin[d][k], n) == 0 and at(gbin_up1[d][k], n) == 1 for d, k in sel], 0.0)
            pc1[n - 1] = joint / norm_of([at(gbin_up1[d][k], n) == 1 for d, k in sel])
        if "condbinlft" in types:
            sel = [c for c in cells if n + 1 <= np_[c[0]][c[1]] and np_up1[c[0]][c[1]] < n + 1]
            joint = mean_or([at(gbin[d][k], n + 1) == 0 and at(gbin[d][k], n) == 1 for d, k in sel], 0.0)
            p_lft[n - 1] = joint / norm_of([at(gbin[d][k], n) == 1 for d, k in sel])
        sel = [c for c in cells if n + np_[c[0]][c[1]] <= L[c[0]][c[1]]]
        p_suf[n - 1] = mean_or([at(gbin[d][k], n + np_[d][k]) == 0 for d, k in sel], 0.0)
        sel = [c for c in cells if n + np_[c[0]][c[1]] <= L[c[0]][c[1]] and n + np_up1[c[0]][c[1]] <= L_up1[c[0]][c[1]]]
        if "conds0" in types:
            joint = mean_or([at(gbin[d][k], n + np_[d][k]) == 0 and at(gbin_up1[d][k], n + np_up1[d][k]) == 0 for d, k in sel], 0.0)
            pcs0[n - 1] = joint / norm_of([at(gbin_up1[d][k], n) == 0 for d, k in sel])  # (the reference norms with y(n): lines 96, 108)
        if "conds1" in types:
            joint = mean_or([at(gbin[d][k], n + np_[d][k]) == 0 and at(gbin_up1[d][k], n + np_up1[d][k]) == 1 for d, k in sel], 0.0)
            pcs1[n - 1] = joint / norm_of([at(gbin_up1[d][k], n) == 1 for d, k in sel])
    n = nlbp + 1
    rest = []
    for d, k in cells:
        if n <= np_[d][k]:
            rest += gbin[d][k][n - 1:np_[d][k]]
    p_rest_pre = mean_or([b == 0 for b in rest], 0.0)
    rest = []
    for d, k in cells:
        if n > np_[d][k] and n <= L[d][k]:
            rest += gbin[d][k][n - 1:]
    p_rest_suf = mean_or([b == 0 for b in rest], 0.0)
    return p_pre + pc0 + pc1 + p_lft + p_suf + pcs0 + pcs1 + [p_rest_pre, p_rest_suf]


def main():
    rng = np.random.default_rng(20171)
    cases = []
    all4 = ["cond0", "cond1", "conds0", "conds1"]
    specs = [  # rows, cols, nq, method, k, types, nlbp, how the symbols are drawn
        (12, 5, 8, "EG", 0, all4, 3, "geom"), (40, 3, 16, "EG", 0, all4 + ["condbinlft"], 3, "geom"),
        (9, 4, 8, "TU", 0, all4, 3, "uniform"), (30, 2, 6, "TU", 0, all4 + ["condbinlft"], 2, "geom"),
        (1, 6, 8, "EG", 0, all4, 3, "uniform"), (25, 1, 32, "EG", 1, all4, 4, "geom"),
        (16, 3, 8, "EG", 0, ["cond0", "conds1"], 3, "zeros"), (20, 4, 4, "TU", 0, all4 + ["condbinlft"], 3, "max"),
    ]
    for rows, cols, nq, method, k, types, nlbp, how in specs:
        if how == "geom":
            G = np.minimum(rng.geometric(0.35, (rows, cols)) - 1, nq - 1)
        elif how == "uniform":
            G = rng.integers(0, nq, (rows, cols))
        elif how == "zeros":
            G = np.zeros((rows, cols), np.int64)
        else:
            G = np.full((rows, cols), nq - 1)
        binarize = (lambda v: tu_code(int(v), nq - 1)) if method == "TU" else (lambda v: eg_code(int(v), k))
        p0 = init_context_model(G.tolist(), binarize, types, nlbp)
        bits = sum(1 << i for i, t in enumerate(["cond0", "cond1", "conds0", "conds1", "condbinlft"]) if t in types)
        cases.append(dict(G=G.tolist(), nq=nq, method=method, k=k, types=bits, nlbp=nlbp, p0=p0))
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "iss_ctx_init_fixture.json")
    json.dump(dict(note="cabacInitContextModel.m restated with Python lists (tests/golden/make_f1_fixture.py); p0 = [p_pre pc0 pc1 "
                        "p_bin_lft p_suf pcs0 pcs1 p_rest_pre p_rest_suf]; types: bit 0 cond0, 1 cond1, 2 conds0, 3 conds1, 4 condbinlft",
                   cases=cases), open(out, "w"))
    print("wrote", out, len(cases), "cases")


if __name__ == "__main__":
    main()
