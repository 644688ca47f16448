#!/usr/bin/env python3
"""Rate of the 16-bit-context batch calls (isscabac_*_batch16_dev: up to 999 contexts, state image in device memory) on
device-resident buffers, for the record in DESIGN.md.  usage: python tools/wide_rate.py [streams] [bins] [contexts]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from isscabac_amd import Engine, stream_bound

S = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
C = int(sys.argv[3]) if len(sys.argv) > 3 else 999
dev = torch.device("cuda:0")
eng = Engine(0)
g = torch.Generator(device=dev).manual_seed(1)
ctx = torch.randint(0, C, (S, B), device=dev, generator=g, dtype=torch.int32).to(torch.int16)
p = torch.rand(C, device=dev, generator=g)
bins = (torch.rand((S, B), device=dev, generator=g) < p[ctx.long()]).to(torch.uint8)
init = np.zeros(C, np.uint8)
cap = S * stream_bound(B)
payload = torch.empty(cap, dtype=torch.uint8, device=dev)
offsets = torch.empty(S + 1, dtype=torch.int64, device=dev)
out = torch.empty((S, B), dtype=torch.uint8, device=dev)
status = torch.empty(S, dtype=torch.int32, device=dev)
torch.cuda.synchronize()


def enc():
    return eng.encode16_dev(bins.data_ptr(), ctx.data_ptr(), S, B, B, init, payload.data_ptr(), cap, offsets.data_ptr())


def dec(total):
    eng.decode16_dev(payload.data_ptr(), total, offsets.data_ptr(), ctx.data_ptr(), S, B, B, init, out.data_ptr(), status.data_ptr())


total = enc()
dec(total)
assert int(status.abs().sum()) == 0 and torch.equal(out, bins)
te, td = [], []
for _ in range(5):
    t0 = time.perf_counter(); total = enc(); t1 = time.perf_counter(); dec(total); t2 = time.perf_counter()
    te.append(t1 - t0); td.append(t2 - t1)
n = S * B
print("%d streams x %d bins, %d contexts (16-bit indices): encode %.2f ms = %.1f Gbin/s, decode %.2f ms = %.1f Gbin/s, round trip %.1f Gbin/s, "
      "%.3f bit/bin" % (S, B, C, min(te) * 1e3, n / min(te) / 1e9, min(td) * 1e3, n / min(td) / 1e9, n / (min(te) + min(td)) / 1e9, 8.0 * total / n))
