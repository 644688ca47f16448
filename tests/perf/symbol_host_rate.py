#!/usr/bin/env python3
"""Config 4 through the HOST form of the symbol API -- what matlab/+coder/cabacEncode.m / cabacDecode.m actually call
(ref: ISS/+coder/cabacEncode.m:23-70, cabacDecode.m:29-55): numpy arrays in pageable host memory in, payload out, PCIe
included.  Three routes, beside the CPU on one core:
  1. C ABI:    isscabac_iss_init_context_model -> isscabac_encode_symbols[_stats] -> isscabac_decode_symbols through ctypes, the
               caller's buffers allocated once
  2. gateway:  the same three steps as MEX commands through tests/mex_gateway.py (matlab/SimpleCABACMex.cpp compiled
               against the stub mex.h): 'issInitContextModel' -> 'encodeSymbols' -> 'decodeSymbols' (up to 81 920 columns)
  3. CPU:      the C restatement of the same walk (binarise, select, code) on a sample of the columns, and the reference's
               own coder classes (oracle/_ref) coding the same bins with the contexts already chosen -- its MATLAB loops
               (one MEX call per bin) are not part of that figure
Shapes: W 400 x 20 and H 109 x 20 (ISS.m:32,36) tiled x 16 384 -> 327 680 columns as independent streams, Nq = 8, EG0,
23 contexts; and config 4 at its real size (40 columns), where a call is launch latency.
    python tests/perf/symbol_host_rate.py [--tile 16384] [--no-gateway] [--real-only]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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


def best_of(fn, n=7 if "--real-only" in sys.argv else 3):
    fn()
    t = []
    for _ in range(n):
        t0 = time.perf_counter()
        out = fn()
        t.append(time.perf_counter() - t0)
    return min(t), out


import ctypes as C
from isscabac_amd import init_by_prob, stream_bound
from isscabac_amd.engine import SymbolStats
L, h = eng._L, eng._h
u8p, u32p, u64p, u16p = C.POINTER(C.c_uint8), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64), C.POINTER(C.c_uint16)
P = lambda a, t: a.ctypes.data_as(t)

shapes = ((400, 20 * tile), (109, 20 * tile), (400, 20 * 1024), (400, 20), (109, 20))
if "--real-only" in sys.argv:  # config 4 at its real size only (a call is launches and host round trips)
    shapes = ((400, 20), (109, 20))
for rows, cols in shapes:
    G = np.minimum(rng.exponential(1.2, (min(cols, 4096), rows)).astype(np.int64), nq - 1).astype(np.uint8)
    sym = np.tile(G, (max(1, cols // 4096), 1))  # [streams, rows]: one column per row of this array
    S = sym.shape[0]
    pitch = (rows + 15) // 16 * 16
    symp = np.zeros((S, pitch), np.uint8)
    symp[:, :rows] = sym
    # 1. the C ABI's host calls with the caller's buffers allocated (and touched) once, as a caller coding matrix after matrix
    #    would hold them -- fresh numpy arrays per call cost more in page faults than the calls themselves
    p0, p0b = eng.iss_init_context_model(sym, params)
    t_init, _ = best_of(lambda: eng.iss_init_context_model(sym, params))
    init = init_by_prob(p0b.astype(np.float64) / 255.0)
    mb = int(L.isscabac_symbol_max_bins(C.byref(params)))
    cap = S * stream_bound(rows * mb)
    payload, offsets, nbins = np.zeros(cap, np.uint8), np.zeros(S + 1, np.uint64), np.zeros(S, np.uint32)
    hist, cost = np.zeros((S, 64), np.uint32), np.zeros((S, 64), np.uint32)
    hb = np.zeros((S, pitch), np.uint16)
    dec, status = np.zeros((S, pitch), np.uint8), np.zeros(S, np.uint32)
    total = C.c_uint64(0)
    st = SymbolStats(P(hist, u32p), P(cost, u32p), P(hb, u16p), None, 0)

    def enc(stats):
        rc = L.isscabac_encode_symbols_stats(h, P(symp, u8p), None, S, rows, pitch, C.byref(params), P(init, u8p), init.size,
                                             P(payload, u8p), cap, P(offsets, u64p), C.byref(total), P(nbins, u32p),
                                             C.byref(st) if stats else None)
        assert rc == 0, eng._L.isscabac_last_error(h)

    def decf():
        rc = L.isscabac_decode_symbols(h, P(payload, u8p), P(offsets, u64p), None, S, rows, pitch, C.byref(params), P(init, u8p),
                                       init.size, P(dec, u8p), P(status, u32p))
        assert rc == 0

    t_enc, _ = best_of(lambda: enc(False))
    t_encs, _ = best_of(lambda: enc(True))
    t_dec, _ = best_of(decf)
    assert (dec[:, :rows] == sym).all() and (status == 0).all()
    nb, tb = int(nbins.sum()), int(total.value)
    print("%7d columns x %3d symbols (%.2f bins/symbol, %5.1f MB in, %5.1f MB payload), C ABI host calls, caller's buffers reused: "
          "init %.2f ms, encode %.2f ms (with the statistics, %.0f MB more out: %.2f ms), decode %.2f ms -> %.2f Gbin/s round trip "
          "(%.2f Gsymbol/s), %.3f ms per round trip"
          % (S, rows, nb / (S * rows), symp.nbytes / 1e6, tb / 1e6, t_init * 1e3, t_enc * 1e3,
             (hist.nbytes + cost.nbytes + hb.nbytes) / 1e6, t_encs * 1e3, t_dec * 1e3, nb / (t_enc + t_dec) / 1e9,
             S * rows / (t_enc + t_dec) / 1e9, (t_enc + t_dec) * 1e3))
    pl = payload[:tb].copy()
    # 2. the MEX gateway (MATLAB shapes: one column per stream; every numeric result a double array, as the reference's are)
    if gw is not None and S <= 20 * 4096:
        symT = np.asfortranarray(sym.T)
        t_gi, (gp0, gp0b) = best_of(lambda: gw.call("issInitContextModel", symT, mex_params, nlhs=2))
        assert (gp0b.ravel() == p0b).all()
        (hh,) = gw.call("initByProb", "/tmp/symrate.bin", gp0b.ravel().astype(np.float64) / 255.0, nlhs=1)
        t_ge, outs = best_of(lambda: gw.call("encodeSymbols", hh, symT, mex_params, nlhs=3))
        gp, go = outs[0], outs[1]
        assert gp.ravel().tobytes() == pl.tobytes()
        t_gd, (gsym, gst) = best_of(lambda: gw.call("decodeSymbols", hh, gp, go, float(rows), mex_params, nlhs=2))
        assert (gsym.T == sym).all() and (gst == 0).all()
        gw.call("delete", hh)
        print("        ... through the MEX gateway (stub mex.h; the python driver's array conversions included): init %.2f ms, encode "
              "%.2f ms, decode %.2f ms -> %.2f Gbin/s round trip" % (t_gi * 1e3, t_ge * 1e3, t_gd * 1e3, nb / (t_ge + t_gd) / 1e9))
    # 3. one host core
    if S > 4096 or rows == 400:
        from oracle import pyoracle
        orc = pyoracle.Oracle()
        n_s = min(S, 2048)
        t0 = time.perf_counter()
        pay = [orc.iss_encode_column(sym[s], nq, "EG", 0, 15, 3, init) for s in range(n_s)]
        for s in range(n_s):
            d, stt = orc.iss_decode_column(pay[s][0], rows, nq, "EG", 0, 15, 3, init)
        t2 = time.perf_counter()
        nbs = sum(p[1] for p in pay)
        print("        ... one host core, C restatement of the whole walk (binarise + select + code), %d columns: %.1f Mbin/s "
              "round trip" % (n_s, nbs / (t2 - t0) / 1e6))
        if pyoracle.have_reference():
            # the reference's coder classes on the same bins with the contexts handed to them (the trace of a GPU encode)
            enc_tr = eng.encode_symbols_stats(sym[:n_s], params, init, trace=True)
            rec, nbn = enc_tr[6], enc_tr[2]
            Bm = int(nbn.min()) // 16 * 16  # the first Bm bins of every column's sequence: equal lengths for the batch call
            r = rec[:, :Bm]
            bins, ctx = np.ascontiguousarray((r >> 10) & 1, np.uint8), np.ascontiguousarray(r & 0x3FF, np.uint8)
            ref = pyoracle.Reference("symrate")
            ref.roundtrip_batch(bins[:64], ctx[:64], init)
            tot, te, td = ref.roundtrip_batch(bins, ctx, init)
            print("        ... one host core, the reference's coder classes on the same bins, contexts given (its MATLAB loops -- one "
                  "MEX call per bin -- not included; one scratch file per stream): %.1f Mbin/s round trip" % (n_s * Bm / (te + td) / 1e6))
