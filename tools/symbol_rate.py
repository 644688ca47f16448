#!/usr/bin/env python3
"""Rate of the symbol-level kernels on BASELINE config 4's shape tiled up: quantised NTF parameter columns (400 rows for
W, 109 for H; ISS.m:32,36) as independent streams, Nq = 8, EG0, ISS context selector with 23 contexts.  Device buffers,
HIP events around the calls (kernel + scan + gather for encode)."""
import ctypes as C
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from isscabac_amd import Engine, SymbolParams, init_by_prob

eng = Engine(0)
L, h = eng._L, eng._h
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
for rows, S in ((400, 20 * 16384), (109, 20 * 16384)):
    pitch = (rows + 15) // 16 * 16
    G = np.minimum(rng.exponential(1.2, (4096, pitch)).astype(np.int64), 7).astype(np.uint8)
    sym = torch.from_numpy(np.tile(G, (S // 4096, 1))).to(dev)
    params = SymbolParams.iss(8, "EG", 0, 15, 3)
    init = init_by_prob(np.full(23, 0.5))
    mb = int(L.isscabac_symbol_max_bins(C.byref(params)))
    cap = S * ((6 * rows * mb + 9 + 7) // 8)
    payload = torch.empty(cap, dtype=torch.uint8, device=dev)
    offsets = torch.empty(S + 1, dtype=torch.int64, device=dev)
    nbins = torch.empty(S, dtype=torch.int32, device=dev)
    out = torch.empty((S, pitch), dtype=torch.uint8, device=dev)
    status = torch.empty(S, dtype=torch.int32, device=dev)
    total = C.c_uint64(0)
    ci = init.ctypes.data_as(C.POINTER(C.c_uint8))
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    for it in range(2):
        ev[0].record()
        rc = L.isscabac_encode_symbols_dev(h, C.c_void_p(sym.data_ptr()), None, S, rows, pitch, C.byref(params), ci, 23,
                                           C.c_void_p(payload.data_ptr()), cap, C.c_void_p(offsets.data_ptr()),
                                           C.byref(total), C.c_void_p(nbins.data_ptr()))
        assert rc == 0
        ev[1].record()
        rc = L.isscabac_decode_symbols_dev(h, C.c_void_p(payload.data_ptr()), total.value, C.c_void_p(offsets.data_ptr()), None,
                                           S, rows, pitch, C.byref(params), ci, 23, C.c_void_p(out.data_ptr()),
                                           C.c_void_p(status.data_ptr()))
        assert rc == 0
        ev[2].record()
        torch.cuda.synchronize()
    assert torch.equal(out[:, :rows], sym[:, :rows]) and int(status.abs().sum()) == 0
    nb = int(nbins.sum())
    e, d = ev[0].elapsed_time(ev[1]), ev[1].elapsed_time(ev[2])
    print("%d streams x %d symbols (%.2f bins/symbol, %.3f bit/bin): encode %.2f ms, decode %.2f ms -> %.1f Gbin/s "
          "(%.1f Gsymbol/s) round trip" % (S, rows, nb / (S * rows), 8.0 * total.value / nb, e, d, nb / (e + d) / 1e6,
                                            S * rows / (e + d) / 1e6))
