#!/usr/bin/env python3
"""The reference's symbol front end restated INDEPENDENTLY of this repository's C code -- plain Python lists standing in
for MATLAB's vectors and cell arrays, statement by statement from the reference's text, 1-based positions kept -- and run
on a few small matrices to produce tests/golden/sym_sel_fixture.json: for every symbol its bin string, for every bin the
context id the selector picks (the id the MEX gateway receives, i.e. MATLAB's ctxID - 1), and what the decoder-side walk
(end-of-symbol rule + de-binariser) makes of the same bins.

The reference functions are MATLAB (no MATLAB in this image), so this second derivation, not a run of the reference, is
what pins rows a14 / a15 of SURVEY 8 beside oracle/cabac_oracle.c's own restatement: the two were written apart, share no
code, and must agree bin for bin (tests/test_sym_fixture.py on CPU; the symbol kernels' bin trace in tests/test_gpu_parity.py).

    binariser            CABAC/cabacBinarizer.m:30-37 (TUCode), :56-69 (EGCode), :71-75 (FLCode)
    de-binariser         CABAC/cabacDebinarizer.m:28-31 (rTUCode), :46-53 (rEGCode), :55-57 (rFLCode)
    end of symbol        CABAC/cabacDecodeSymbolFinished.m:12-30
    ISS selector         ISS/+coder/cabacContextSelection.m:24-67
    its two callers      ISS/+coder/cabacEncode.m:45-70, ISS/+coder/cabacDecode.m:29-55
    demo selector        CABAC/cabacDemo.m:113-121 (encoder), :160-169 (decoder)
"""
import json
import math
import os
import random

TYPE_BITS = {"cond0": 1, "cond1": 2, "conds0": 4, "conds1": 8, "condbinlft": 16}  # include/isscabac.h: iss_types


# ---- cabacBinarizer.m ---------------------------------------------------------------------------------------------------
def FLCode(v, nbits):  # :71-75   i=1:nbits; l = floor(v ./ 2.^(nbits-i)); C = rem(l,2);
    return [int(math.floor(v / 2 ** (nbits - i))) % 2 for i in range(1, nbits + 1)]


def TUCode(v, maxVal):  # :30-37
    if v == maxVal:
        return [1] * maxVal
    C = [1] * (v + 1)
    C[(v + 1) - 1] = 0
    return C


def EGCode(v, k):  # :56-69
    n_p = int(math.floor(math.log2((v / 2 ** k) + 1))) + 1
    n_s = k + n_p - 1
    C = [1] * (n_p + n_s)
    C[n_p - 1] = 0
    v_s = v - 2 ** k * (2 ** (n_p - 1) - 1)
    C[n_p:] = FLCode(v_s, n_s)  # C(n_p+1:end)
    return C


def binarize(v, Nq, method):  # :10-28
    maxVal = Nq - 1
    if method == "DEC2TU":
        return TUCode(v, maxVal)
    assert method in ("DEC2EG0", "DEC2EG1", "DEC2EG2")
    return EGCode(v, int(method[-1]))


# ---- cabacDebinarizer.m --------------------------------------------------------------------------------------------------
def find_first_zero(c):  # find(~c,1,'first'): 1-based position of the first 0, None for "empty"
    for i, b in enumerate(c):
        if b == 0:
            return i + 1
    return None


def rFLCode(c, nbits):  # :55-57   v = c*2.^((nbits-1):-1:0)'
    assert len(c) == nbits
    return sum(b * 2 ** e for b, e in zip(c, range(nbits - 1, -1, -1)))


def debinarize(c, Nq, method):  # :11-25
    maxVal = Nq - 1
    if method == "DEC2TU":  # rTUCode :28-31
        v = find_first_zero(c)
        return maxVal if v is None else v - 1
    k = int(method[-1])  # rEGCode :46-53
    n_p = find_first_zero(c)
    xi = c[n_p:]  # xi = c; xi(1:n_p) = []
    n_s = len(xi)
    return 2 ** k * (2 ** (n_p - 1) - 1) + rFLCode(xi, n_s)


# ---- cabacDecodeSymbolFinished.m -----------------------------------------------------------------------------------------
def symbol_finished(g, n, Nq, binMethod, n_p, n_s):  # :12-30; g(n) is g[n-1]
    isFinished = False
    if binMethod == "DEC2TU":
        isFinished = g[n - 1] == 0 or n == Nq - 1
    else:
        egk = int(binMethod[-1])
        if n_s == -1:  # decode prefix
            if g[n - 1] == 0:
                n_p = n_p + n
                n_s = egk + n_p - 1
                if n_s == 0:
                    isFinished = True
        else:  # decode suffix
            if n_s == 1:
                isFinished = True
            else:
                n_s = n_s - 1
    return isFinished, n, n_p, n_s


# ---- +coder/cabacContextSelection.m --------------------------------------------------------------------------------------
def context_selection(n, g_prev, g_up1, types, Nlbp):  # :24-67; returns MATLAB's 1-based ctxID
    np_prev = find_first_zero(g_prev)
    isPrefix_prev = True
    if np_prev is not None and n > np_prev:
        isPrefix_prev = False
    np_up1 = find_first_zero(g_up1)
    isPrefix_up1 = True
    if np_up1 is not None and n > np_up1:
        isPrefix_up1 = False

    if isPrefix_prev:  # we are in the prefix
        if n <= Nlbp:
            ctxID = n
            if len(g_up1) >= n and isPrefix_up1:
                if g_up1[n - 1] == 0 and "cond0" in types:
                    ctxID = Nlbp + n
                elif g_up1[n - 1] == 1 and "cond1" in types:
                    ctxID = 2 * Nlbp + n
            elif n > 1 and g_prev[n - 2] == 1 and "condbinlft" in types:
                ctxID = 3 * Nlbp + n - 1
        else:
            ctxID = 7 * Nlbp + 1
    else:  # now we are in the suffix
        if n - np_prev <= Nlbp:
            ctxID = 4 * Nlbp + n - np_prev
            if len(g_up1) >= n and not isPrefix_up1:
                if g_up1[n - 1] == 0 and "conds0" in types:
                    ctxID = 5 * Nlbp + n - np_prev
                elif g_up1[n - 1] == 1 and "conds1" in types:
                    ctxID = 6 * Nlbp + n - np_prev
        else:
            ctxID = 7 * Nlbp + 2
    return ctxID


# ---- +coder/cabacEncode.m:45-70: the walk over one matrix (rows d, columns k) ------------------------------------------------
def iss_encode_walk(G, Nq, method, types, Nlbp):
    """-> per column k: list over rows d of (bin string, [ctxID - 1 per bin])"""
    D, K = len(G), len(G[0])
    Gbin = [[binarize(G[d][k], Nq, method) for k in range(K)] for d in range(D)]
    cols = []
    for k in range(K):
        col = []
        for d in range(D):
            g = Gbin[d][k]
            g_up1 = Gbin[d - 1][k] if d > 0 else []  # if d>1, g_up1 = Gbin{d-1,k}; else, g_up1=[]
            ids = []
            for cnt in range(1, len(g) + 1):
                ctxID = context_selection(cnt, g[:cnt - 1], g_up1, types, Nlbp)
                ids.append(ctxID - 1)  # c.encodeBin(g(cnt),ctxID-1)
            col.append((g, ids))
        cols.append(col)
    return cols


# ---- +coder/cabacDecode.m:29-55 with the bins handed in instead of decoded ---------------------------------------------------
def iss_decode_walk(bins_of_column, D, Nq, method, types, Nlbp):
    """bins_of_column: the column's bins back to back.  -> (symbols, [ctxID - 1 per bin], bins consumed)"""
    pos, Gbin, ids = 0, [], []
    for d in range(D):
        g_up1 = Gbin[d - 1] if d > 0 else []
        g, n, isFinished, n_s, n_p = [], 1, False, -1, 0
        while not isFinished:
            ctxID = context_selection(n, g[:n - 1], g_up1, types, Nlbp)
            ids.append(ctxID - 1)
            g.append(bins_of_column[pos])  # g(n) = c.decodeBin(ctxID-1)
            pos += 1
            isFinished, n, n_p, n_s = symbol_finished(g, n, Nq, method, n_p, n_s)
            n = n + 1
        Gbin.append(g)
    return [debinarize(g, Nq, method) for g in Gbin], ids, pos


# ---- cabacDemo.m:96-126 (encoder) and :150-175 (decoder) ------------------------------------------------------------------
def demo_select(n, g_lft):  # :113-121 / :160-169, param.disableContexts = false
    if n == 1 and len(g_lft) > 0:
        if g_lft[0] == 1:
            return 1
        if g_lft[0] == 0:
            return 2
    return 0


def demo_encode_walk(symbols, Nq):
    strings = [binarize(v, Nq, "DEC2TU") for v in symbols]
    out = []
    for it, g in enumerate(strings):
        g_lft = strings[it - 1] if it > 0 else []
        out.append((g, [demo_select(n, g_lft) for n in range(1, len(g) + 1)]))
    return out


def demo_decode_walk(bins, nsym, Nq):
    pos, strings, ids = 0, [], []
    for it in range(nsym):
        g_lft = strings[it - 1] if it > 0 else []
        g, n, isFinished, n_s, n_p = [], 1, False, -1, 0
        while not isFinished:
            ids.append(demo_select(n, g_lft))
            g.append(bins[pos])
            pos += 1
            isFinished, n, n_p, n_s = symbol_finished(g, n, Nq, "DEC2TU", n_p, n_s)
            n = n + 1
        strings.append(g)
    return [debinarize(g, Nq, "DEC2TU") for g in strings], ids, pos


# ---- the cases ------------------------------------------------------------------------------------------------------------
def matrix(rng, D, K, Nq, zero_heavy):
    def draw():
        if zero_heavy:
            return min(Nq - 1, int(abs(rng.gauss(0.0, Nq / 4.0))))
        return rng.randrange(Nq)
    return [[draw() for _ in range(K)] for _ in range(D)]


def main():
    rng = random.Random(20260404)
    all_types = ["cond0", "cond1", "conds0", "conds1"]
    iss_cases = [
        # name, D, K, Nq, method, types, Nlbp, zero-heavy values
        ("iss_default_eg0_nq8", 14, 3, 8, "DEC2EG0", all_types, 3, True),          # ISS/ISS.m:50-52
        ("iss_eg0_nq16_long_prefix_and_suffix", 12, 2, 16, "DEC2EG0", all_types, 3, False),  # n > Nlbp on both sides
        ("iss_eg1_nq12", 10, 2, 12, "DEC2EG1", all_types, 3, False),
        ("iss_eg2_nq24", 10, 2, 24, "DEC2EG2", all_types, 2, False),
        ("iss_tu_nq5", 12, 2, 5, "DEC2TU", all_types, 3, False),
        ("iss_tu_nq9_rest_context", 12, 2, 9, "DEC2TU", all_types, 3, False),       # bins 4..8 take 7*Nlbp+1
        ("iss_types_none", 10, 2, 8, "DEC2EG0", [], 3, True),
        ("iss_types_cond0_only", 10, 2, 8, "DEC2EG0", ["cond0"], 3, True),
        ("iss_types_cond1_only", 10, 2, 8, "DEC2EG0", ["cond1"], 3, True),
        ("iss_types_conds0_only", 10, 2, 8, "DEC2EG0", ["conds0"], 3, False),
        ("iss_types_conds1_only", 10, 2, 8, "DEC2EG0", ["conds1"], 3, False),
        ("iss_types_condbinlft_only", 10, 2, 8, "DEC2EG0", ["condbinlft"], 3, False),
        ("iss_types_all_five", 12, 2, 8, "DEC2EG0", all_types + ["condbinlft"], 3, False),
        ("iss_condbinlft_tu", 10, 2, 7, "DEC2TU", ["cond1", "condbinlft"], 3, False),
        ("iss_nlbp1", 10, 2, 8, "DEC2EG0", all_types, 1, False),
        ("iss_nlbp4_nq32", 10, 2, 32, "DEC2EG0", all_types, 4, False),
        ("iss_one_row", 1, 5, 8, "DEC2EG0", all_types, 3, False),                   # no symbol one row up anywhere
        ("iss_one_column_all_zero", 9, 1, 8, "DEC2EG0", all_types, 3, None),
        ("iss_maximum_values_tu", 6, 2, 6, "DEC2TU", all_types, 3, "max"),          # TU strings without a terminating 0
    ]
    out = {"note": "generated by tests/golden/make_sym_fixture.py: an independent Python restatement of the reference's "
                   "MATLAB text (no code shared with oracle/ or the kernels); ctx = the id handed to the gateway "
                   "(MATLAB's ctxID - 1)",
           "iss": [], "demo": [], "binarizer": {}}
    for name, D, K, Nq, method, types, Nlbp, kind in iss_cases:
        if kind is None:
            G = [[0] * K for _ in range(D)]
        elif kind == "max":
            G = [[(Nq - 1) if (d + k) % 3 else rng.randrange(Nq) for k in range(K)] for d in range(D)]
        else:
            G = matrix(rng, D, K, Nq, kind)
        cols = iss_encode_walk(G, Nq, method, types, Nlbp)
        case = {"name": name, "rows": D, "cols": K, "nq": Nq, "method": method, "types": types,
                "types_mask": sum(TYPE_BITS[t] for t in types), "nlbp": Nlbp, "G": G, "columns": []}
        for k, col in enumerate(cols):
            bins = [b for g, _ in col for b in g]
            ids = [c for _, i in col for c in i]
            assert all(0 <= c < 7 * Nlbp + 2 for c in ids)
            # the decoder's walk over the same bins finds the same contexts, the same symbol boundaries and the symbols
            sym, ids_dec, used = iss_decode_walk(bins, D, Nq, method, types, Nlbp)
            assert sym == [G[d][k] for d in range(D)] and ids_dec == ids and used == len(bins), name
            case["columns"].append({"bins_per_symbol": [len(g) for g, _ in col], "bins": bins, "ctx": ids})
        out["iss"].append(case)
    for name, n, Nq in (("demo_nq4", 40, 4), ("demo_nq2", 24, 2), ("demo_nq9", 30, 9), ("demo_one_symbol", 1, 4)):
        symbols = [min(Nq - 1, int(abs(rng.gauss(0.0, Nq / 3.0)))) for _ in range(n)]  # cabacDemo.m:11-37: |randn|-like
        walk = demo_encode_walk(symbols, Nq)
        bins = [b for g, _ in walk for b in g]
        ids = [c for _, i in walk for c in i]
        sym, ids_dec, used = demo_decode_walk(bins, n, Nq)
        assert sym == symbols and ids_dec == ids and used == len(bins), name
        out["demo"].append({"name": name, "nq": Nq, "symbols": symbols, "bins_per_symbol": [len(g) for g, _ in walk],
                            "bins": bins, "ctx": ids})
    # every bin string of the four binarisers for small alphabets, and the de-binariser on each
    for method, Nq in (("DEC2TU", 2), ("DEC2TU", 8), ("DEC2TU", 33), ("DEC2EG0", 64), ("DEC2EG1", 64), ("DEC2EG2", 64)):
        tab = []
        for v in range(Nq):
            c = binarize(v, Nq, method)
            assert debinarize(c, Nq, method) == v
            tab.append(c)
        out["binarizer"]["%s_nq%d" % (method, Nq)] = tab
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "sym_sel_fixture.json")
    with open(path, "w") as f:
        json.dump(out, f, separators=(",", ":"))
    nb = sum(len(c["bins"]) for case in out["iss"] for c in case["columns"]) + sum(len(c["bins"]) for c in out["demo"])
    print("%s: %d ISS cases, %d demo cases, %d bins, %d bytes" % (path, len(out["iss"]), len(out["demo"]), nb, os.path.getsize(path)))


if __name__ == "__main__":
    main()
