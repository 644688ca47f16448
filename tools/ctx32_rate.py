#!/usr/bin/env python3
"""Kernel-only round-trip rate with 32 contexts (occupancy experiments with a smaller state image: ISSCABAC_EXP_SB32)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from isscabac_amd import Engine
eng = Engine(0)
dev = torch.device("cuda:0")
S, B, C = 1 << 20, 4096, 32
bins = torch.empty((S, B), dtype=torch.uint8, device=dev); ctx = torch.empty((S, B), dtype=torch.uint8, device=dev)
cap = S * 3104  # (the slot bound: timing-only ablation builds emit garbage of any size)
payload = torch.empty(cap, dtype=torch.uint8, device=dev); offsets = torch.empty(S + 1, dtype=torch.int64, device=dev)
out = torch.empty((S, B), dtype=torch.uint8, device=dev); status = torch.empty((S,), dtype=torch.int32, device=dev)
eng.set_profiling(True)
eng.synth_dev(0xCABAC003, 0, S, B, B, 64, False, bins.data_ptr(), ctx.data_ptr())
ctx.bitwise_and_(31)
init = np.zeros(C, np.uint8)
es, ds = [], []
for _ in range(6):
    total = eng.encode_dev(bins.data_ptr(), ctx.data_ptr(), S, B, B, init, payload.data_ptr(), cap, offsets.data_ptr())
    es.append(eng.last_kernel_ms()["encode"])
    eng.decode_dev(payload.data_ptr(), total, offsets.data_ptr(), ctx.data_ptr(), S, B, B, init, out.data_ptr(), status.data_ptr())
    ds.append(eng.last_kernel_ms()["decode"])
assert os.environ.get('ISSCABAC_NO_VERIFY') or (torch.equal(out, bins) and int(status.abs().sum()) == 0)
print("32 contexts: encode %.3f ms decode %.3f ms (medians of 5)" % (sorted(es[1:])[2], sorted(ds[1:])[2]))
