#!/usr/bin/env python3
"""isscabac_iss_init_context_model_dev (the empirical context initialisation of cabacInitContextModel.m) on
device-resident symbols: config 4's two shapes tiled x 16 384.  usage: python tools/init_rate.py"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from isscabac_amd import Engine, SymbolParams

eng = Engine(0)
L, h = eng._L, eng._h
L.isscabac_iss_init_context_model_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.POINTER(SymbolParams),
                                                  C.POINTER(C.c_double), C.POINTER(C.c_uint8)]
rng = np.random.default_rng(0)
params = SymbolParams.iss(8, "EG", 0, 15, 3)
for rows, S in ((400, 20 * 16384), (109, 20 * 16384), (400, 20)):
    pitch = (rows + 15) // 16 * 16
    G = np.minimum(rng.exponential(1.2, (min(S, 4096), pitch)).astype(np.int64), 7).astype(np.uint8)
    sym = torch.from_numpy(np.tile(G, (max(1, S // 4096), 1))).cuda()
    p0 = (C.c_double * 23)()
    b0 = (C.c_uint8 * 23)()
    t = []
    for it in range(6):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rc = L.isscabac_iss_init_context_model_dev(h, C.c_void_p(sym.data_ptr()), S, rows, pitch, C.byref(params), p0, b0)
        t.append(time.perf_counter() - t0)
        assert rc == 0
    want_p0, want_b = eng.iss_init_context_model(sym.cpu().numpy()[:, :rows], params)
    assert list(b0) == want_b.tolist()
    print("%7d columns x %3d rows: %.3f ms per call (%.1f Gsymbol/s), device-resident symbols, result on the host" % (S, rows, min(t[1:]) * 1e3, S * rows / min(t[1:]) / 1e9))
