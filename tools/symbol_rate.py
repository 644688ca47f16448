#!/usr/bin/env python3
"""Rate of the symbol-level kernels on BASELINE config 4's shape tiled up: quantised NTF parameter columns (400 rows for
W, 109 for H; ISS.m:32,36) as independent streams, Nq = 8, EG0, ISS context selector with 23 contexts.  Device buffers,
HIP events around the ASYNCHRONOUS entry points (isscabac_*_symbols_dev_async: kernel + scan + gather for encode; the trie kernel
and its fallback kernel for decode), several steps chained with no host wait in between, like bench.py's steps; the synchronous
calls (one host round trip each) are timed beside them."""
import ctypes as C
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from isscabac_amd import Engine, SymbolParams, init_by_prob

eng = Engine(0)
L, h = eng._L, eng._h
dev = torch.device("cuda:0")
ts = torch.cuda.Stream(dev)  # the engine launches on this stream, so that torch's events bracket its kernels
eng.set_stream(ts.cuda_stream)
torch.cuda.set_stream(ts)
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
    for it in range(2):  # the synchronous calls: each returns after its own stream synchronisation
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
    e_sync, d_sync = ev[0].elapsed_time(ev[1]), ev[1].elapsed_time(ev[2])
    out.zero_()
    steps = 10
    evs = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(steps)]
    for it in range(steps):  # encode -> decode -> encode ... only enqueued; one synchronisation at the end
        evs[it][0].record()
        rc = L.isscabac_encode_symbols_dev_async(h, C.c_void_p(sym.data_ptr()), None, S, rows, pitch, C.byref(params), ci, 23,
                                                 C.c_void_p(payload.data_ptr()), cap, C.c_void_p(offsets.data_ptr()),
                                                 C.c_void_p(nbins.data_ptr()))
        assert rc == 0
        evs[it][1].record()
        rc = L.isscabac_decode_symbols_dev_async(h, C.c_void_p(payload.data_ptr()), cap, C.c_void_p(offsets.data_ptr()), None, S, rows,
                                                 pitch, C.byref(params), ci, 23, C.c_void_p(out.data_ptr()), C.c_void_p(status.data_ptr()))
        assert rc == 0
        evs[it][2].record()
    tot2 = C.c_uint64(0)
    assert L.isscabac_sync(h, C.byref(tot2)) == 0 and tot2.value == total.value
    torch.cuda.synchronize()
    assert torch.equal(out[:, :rows], sym[:, :rows]) and int(status.abs().sum()) == 0
    e = sorted(x[0].elapsed_time(x[1]) for x in evs[2:])[len(evs[2:]) // 2]
    d = sorted(x[1].elapsed_time(x[2]) for x in evs[2:])[len(evs[2:]) // 2]
    print("%d streams x %d symbols (%.2f bins/symbol, %.3f bit/bin): encode %.3f ms, decode %.3f ms -> %.1f Gbin/s "
          "(%.1f Gsymbol/s) round trip   [asynchronous entry points, medians of %d chained steps; the synchronous calls, a host round "
          "trip each: %.2f + %.2f ms = %.1f Gbin/s]"
          % (S, rows, nb / (S * rows), 8.0 * total.value / nb, e, d, nb / (e + d) / 1e6, S * rows / (e + d) / 1e6, steps - 2,
             e_sync, d_sync, nb / (e_sync + d_sync) / 1e6))
