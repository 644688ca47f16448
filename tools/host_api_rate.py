#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer entry points (isscabac_encode_batch / isscabac_decode_batch): numpy arrays in
pageable host memory in, payload out.  Reported in DESIGN.md section 6; never bench.py's `value`."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401
from isscabac_amd import Engine

S, B, C = 1 << 16, 4096, 64
rng = np.random.default_rng(0)
ctx = rng.integers(0, C, (S, B), dtype=np.uint8)
bins = (rng.random((S, B)) < (2 * ctx + 1) / (2.0 * C)).astype(np.uint8)
init = np.zeros(C, np.uint8)
eng = Engine(0)
eng.encode(bins[:256], ctx[:256], init)
for _ in range(2):
    t0 = time.perf_counter()
    payload, offsets = eng.encode(bins, ctx, init)
    t1 = time.perf_counter()
    dec, status = eng.decode(payload, offsets, ctx, init)
    t2 = time.perf_counter()
assert (dec == bins).all() and (status == 0).all()
n = S * B
print("host-buffer API, %d streams x %d bins (%.0f MB in): encode %.1f ms = %.2f Gbin/s, decode %.1f ms = %.2f Gbin/s, "
      "round trip %.2f Gbin/s" % (S, B, 2 * n / 1e6, (t1 - t0) * 1e3, n / (t1 - t0) / 1e9, (t2 - t1) * 1e3, n / (t2 - t1) / 1e9,
                                  n / (t2 - t0) / 1e9))
