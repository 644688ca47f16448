#!/usr/bin/env python3
"""Kernel-only round-trip rate for different context counts (which select the <= 64 / <= 128 / <= 255 context kernels)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from isscabac_amd import Engine
eng = Engine(0)
dev = torch.device("cuda:0")
S, B = 1 << 20, 4096
bins = torch.empty((S, B), dtype=torch.uint8, device=dev); ctx = torch.empty((S, B), dtype=torch.uint8, device=dev)
cap = S * 1100
payload = torch.empty(cap, dtype=torch.uint8, device=dev); offsets = torch.empty(S + 1, dtype=torch.int64, device=dev)
out = torch.empty((S, B), dtype=torch.uint8, device=dev); status = torch.empty((S,), dtype=torch.int32, device=dev)
eng.set_profiling(True)
if len(sys.argv) > 1:
    eng.set_tuning(sys.argv[1])  # throughput | latency: force the kernel instance
for C, mul in ((3, 1), (4, 1), (64, 1), (127, 2), (190, 3), (253, 4)):
    eng.synth_dev(0xCABAC003, 0, S, B, B, 3 if C == 3 else (64 if C > 4 else 64), False, bins.data_ptr(), ctx.data_ptr())
    if C == 4:
        ctx.bitwise_and_(3)
    if mul != 1:
        ctx.mul_(mul)
    init = np.zeros(C, np.uint8)
    for _ in range(2):
        total = eng.encode_dev(bins.data_ptr(), ctx.data_ptr(), S, B, B, init, payload.data_ptr(), cap, offsets.data_ptr())
        e = eng.last_kernel_ms()["encode"]
        eng.decode_dev(payload.data_ptr(), total, offsets.data_ptr(), ctx.data_ptr(), S, B, B, init, out.data_ptr(), status.data_ptr())
        d = eng.last_kernel_ms()["decode"]
    assert torch.equal(out, bins) and int(status.abs().sum()) == 0
    print("%3d contexts: encode %.3f ms decode %.3f ms -> %.1f Gbin/s round trip (kernels only), %.3f bit/bin" % (
        C, e, d, S * B / (e + d) / 1e6, 8.0 * total / (S * B)))
