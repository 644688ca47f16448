#!/usr/bin/env python3
"""Kernel-only rate with per-stream bin counts: uniform in [lo*B, B] (how much the ragged tail costs)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from isscabac_amd import Engine
eng = Engine(0)
dev = torch.device("cuda:0")
S, B, C = 1 << 20, 4096, 64
bins = torch.empty((S, B), dtype=torch.uint8, device=dev); ctx = torch.empty((S, B), dtype=torch.uint8, device=dev)
eng.synth_dev(0xCABAC003, 0, S, B, B, C, False, bins.data_ptr(), ctx.data_ptr())
cap = S * 1100
payload = torch.empty(cap, dtype=torch.uint8, device=dev); offsets = torch.empty(S + 1, dtype=torch.int64, device=dev)
out = torch.empty((S, B), dtype=torch.uint8, device=dev); status = torch.empty((S,), dtype=torch.int32, device=dev)
init = np.zeros(C, np.uint8)
eng.set_profiling(True)
for lo in (1.0, 0.9, 0.5, 0.0):
    nb = torch.randint(int(lo * B), B + 1, (S,), dtype=torch.int32, device=dev)
    for _ in range(2):
        total = eng.encode_dev(bins.data_ptr(), ctx.data_ptr(), S, B, B, init, payload.data_ptr(), cap, offsets.data_ptr(),
                               d_nbins=nb.data_ptr())
        e = eng.last_kernel_ms()["encode"]
        eng.decode_dev(payload.data_ptr(), total, offsets.data_ptr(), ctx.data_ptr(), S, B, B, init, out.data_ptr(),
                       status.data_ptr(), d_nbins=nb.data_ptr())
        d = eng.last_kernel_ms()["decode"]
    nbins = int(nb.sum())
    mask = torch.arange(B, device=dev)[None, :] < nb[:, None]
    assert int(status.abs().sum()) == 0 and torch.equal(out * mask, bins * mask)
    print("bins per stream in [%4d, %d]: encode %.2f ms decode %.2f ms -> %.1f Gbin/s round trip (kernels only)" % (
        int(lo * B), B, e, d, nbins / (e + d) / 1e6))
