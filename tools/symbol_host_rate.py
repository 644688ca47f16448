#!/usr/bin/env python3
"""Config 4 through the HOST form of the symbol API -- what matlab/+coder/cabacEncode.m / cabacDecode.m actually call
(ref: ISS/+coder/cabacEncode.m:23-70, cabacDecode.m:29-55): numpy arrays in pageable host memory in, payload out, PCIe
included.  Three routes, beside the CPU on one core:
  1. ctypes:   isscabac_iss_init_context_model -> isscabac_encode_symbols_stats -> isscabac_decode_symbols
  2. gateway:  the same three steps as MEX commands through tests/mex_gateway.py (matlab/SimpleCABACMex.cpp compiled
               against the stub mex.h): 'issInitContextModel' -> 'encodeSymbolsStats' -> 'decodeSymbols'
  3. CPU:      the C restatement of the same walk (binarise, select, code) on a sample of the columns, and the reference's
               own coder classes (oracle/_ref) coding the same bins with the contexts already chosen -- its MATLAB loops
               (one MEX call per bin) are not part of that figure
Shapes: W 400 x 20 and H 109 x 20 (ISS.m:32,36) tiled x 16 384 -> 327 680 columns as independent streams, Nq = 8, EG0,
23 contexts; and config 4 at its real size (40 columns), where a call is launch latency.
    python tools/symbol_host_rate.py [--tile 16384] [--no-gateway]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch  # noqa: F401
from isscabac_amd import Engine, SymbolParams

tile = 16384
if "--tile" in sys.argv:
    tile = int(sys.argv[sys.argv.index("--tile") + 1])
use_gw = "--no-gateway" not in sys.argv
eng = Engine(0)
rng = np.random.default_rng(0)
nq = 8
params = SymbolParams.iss(nq, "EG", 0, 15, 3)
gw = None
if use_gw:
    from mex_gateway import Gateway
    gw = Gateway()
mex_params = np.array([1.0, 0.0, nq, 1.0, 15.0, 3.0, 0.0])


def best_of(fn, n=3):
    fn()
    t = []
    for _ in range(n):
        t0 = time.perf_counter()
        out = fn()
        t.append(time.perf_counter() - t0)
    return min(t), out


for rows, cols in ((400, 20 * tile), (109, 20 * tile), (400, 20), (109, 20)):
    G = np.minimum(rng.exponential(1.2, (min(cols, 4096), rows)).astype(np.int64), nq - 1).astype(np.uint8)
    sym = np.tile(G, (max(1, cols // 4096), 1))  # [streams, rows]: one column per row of this array
    S = sym.shape[0]
    # 1. ctypes host calls
    t_init, (p0, p0b) = best_of(lambda: eng.iss_init_context_model(sym, params))
    from isscabac_amd import init_by_prob
    init = init_by_prob(p0b.astype(np.float64) / 255.0)
    t_enc, enc = best_of(lambda: eng.encode_symbols_stats(sym, params, init))
    payload, offsets, nbins = enc[0], enc[1], enc[2]
    t_dec, (dec, status) = best_of(lambda: eng.decode_symbols(payload, offsets, rows, params, init))
    assert (dec == sym).all() and (status == 0).all()
    nb = int(nbins.sum())
    print("%7d columns x %3d symbols (%.2f bins/symbol, %5.1f MB in, %5.1f MB payload), ctypes host calls: init %.2f ms, "
          "encode+stats %.2f ms, decode %.2f ms -> %.2f Gbin/s round trip (%.2f Gsymbol/s), %.3f ms per round trip"
          % (S, rows, nb / (S * rows), sym.nbytes / 1e6, payload.size / 1e6, t_init * 1e3, t_enc * 1e3, t_dec * 1e3,
             nb / (t_enc + t_dec) / 1e9, S * rows / (t_enc + t_dec) / 1e9, (t_enc + t_dec) * 1e3))
    # 2. the MEX gateway (MATLAB shapes: one column per stream)
    if gw is not None:
        symT = np.asfortranarray(sym.T)
        t_gi, (gp0, gp0b) = best_of(lambda: gw.call("issInitContextModel", symT, mex_params, nlhs=2))
        assert (gp0b.ravel() == p0b).all()
        (h,) = gw.call("initByProb", "/tmp/symrate.bin", gp0b.ravel().astype(np.float64) / 255.0, nlhs=1)
        t_ge, outs = best_of(lambda: gw.call("encodeSymbolsStats", h, symT, mex_params, nlhs=6))
        gp, go = outs[0], outs[1]
        assert gp.ravel().tobytes() == payload.tobytes()
        t_gd, (gsym, gst) = best_of(lambda: gw.call("decodeSymbols", h, gp, go, float(rows), mex_params, nlhs=2))
        assert (gsym.T == sym).all() and (gst == 0).all()
        gw.call("delete", h)
        print("        ... through the MEX gateway (stub mex.h: array conversions included): init %.2f ms, encode+stats %.2f ms, "
              "decode %.2f ms -> %.2f Gbin/s round trip" % (t_gi * 1e3, t_ge * 1e3, t_gd * 1e3, nb / (t_ge + t_gd) / 1e9))
    # 3. one host core
    if S > 4096 or rows == 400:
        from oracle import pyoracle
        orc = pyoracle.Oracle()
        n_s = min(S, 2048)
        t0 = time.perf_counter()
        pay = [orc.iss_encode_column(sym[s], nq, "EG", 0, 15, 3, init) for s in range(n_s)]
        t1 = time.perf_counter()
        for s in range(n_s):
            d, st = orc.iss_decode_column(pay[s][0], rows, nq, "EG", 0, 15, 3, init)
        t2 = time.perf_counter()
        nbs = sum(p[1] for p in pay)
        print("        ... one host core, C restatement of the whole walk (binarise + select + code), %d columns: %.1f Mbin/s "
              "round trip" % (n_s, nbs / (t2 - t0) / 1e6))
        if pyoracle.have_reference():
            # the reference's coder classes on the same bins with the contexts handed to them (the trace of the encode above)
            enc_tr = eng.encode_symbols_stats(sym[:n_s], params, init, trace=True)
            rec, nbn = enc_tr[6], enc_tr[2]
            ref = pyoracle.Reference("symrate")
            B = int(nbn.max())
            Bp = (B + 15) // 16 * 16
            bins = np.zeros((n_s, Bp), np.uint8)
            ctx = np.zeros((n_s, Bp), np.uint8)
            for s in range(n_s):
                r = rec[s, :nbn[s]]
                bins[s, :nbn[s]], ctx[s, :nbn[s]] = (r >> 10) & 1, r & 0x3FF
            tot = te = td = 0.0
            for s in range(n_s):
                p = ref.encode_stream(bins[s, :nbn[s]], ctx[s, :nbn[s]], init)
            t0 = time.perf_counter()
            for s in range(n_s):
                p = ref.encode_stream(bins[s, :nbn[s]], ctx[s, :nbn[s]], init)
                d = ref.decode_stream(p, ctx[s, :nbn[s]], init)
            t1 = time.perf_counter()
            print("        ... one host core, the reference's coder classes on the same bins, contexts given (one scratch file per "
                  "stream): %.1f Mbin/s round trip" % (int(nbn.sum()) / (t1 - t0) / 1e6))
