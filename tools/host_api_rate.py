#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer entry points (isscabac_encode_batch / isscabac_decode_batch): numpy arrays in
pageable host memory in, payload out -- what a MEX / FFI caller gets.  Reported in DESIGN.md section 6; never bench.py's
`value`.  Outputs are reused between calls (the `out=` buffers), as a caller that codes batch after batch would."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401
from isscabac_amd import Engine, stream_bound

S, B, C = 1 << 16, 4096, 64
if len(sys.argv) > 1:
    S = int(sys.argv[1])
rng = np.random.default_rng(0)
ctx = rng.integers(0, C, (S, B), dtype=np.uint8)
bins = (rng.random((S, B)) < (2 * ctx + 1) / (2.0 * C)).astype(np.uint8)
init = np.zeros(C, np.uint8)
eng = Engine(0)
eng.encode(bins[:256], ctx[:256], init)
pbuf, obuf = np.zeros(S * stream_bound(B), np.uint8), np.zeros(S + 1, np.uint64)
dbuf, sbuf = np.zeros((S, B), np.uint8), np.zeros(S, np.uint32)
best = [1e9, 1e9]
for it in range(4):
    t0 = time.perf_counter()
    payload, offsets = eng.encode(bins, ctx, init, out=(pbuf, obuf))
    t1 = time.perf_counter()
    dec, status = eng.decode(payload, offsets, ctx, init, out=(dbuf, sbuf))
    t2 = time.perf_counter()
    if it:
        best = [min(best[0], t1 - t0), min(best[1], t2 - t1)]
assert (dec == bins).all() and (status == 0).all()
n = S * B
te, td = best
print("host-buffer API, %d streams x %d bins (%.0f MB in): encode %.1f ms = %.2f Gbin/s (%.1f GB/s over the link), decode %.1f ms = "
      "%.2f Gbin/s, round trip %.2f Gbin/s (best of 3 after a warm-up call)"
      % (S, B, 2 * n / 1e6, te * 1e3, n / te / 1e9, 2 * n / te / 1e9, td * 1e3, n / td / 1e9, n / (te + td) / 1e9))

# ... and with the bins as bits on the host side (isscabac_*_batch_packed): byte-exact against the calls above
bits = np.packbits(bins, axis=1, bitorder="little")
bbuf = np.zeros((S, B // 8), np.uint8)
best = [1e9, 1e9]
for it in range(4):
    t0 = time.perf_counter()
    payload2, offsets2 = eng.encode_packed(bits, ctx, init, out=(pbuf, obuf))
    t1 = time.perf_counter()
    dbits, status = eng.decode_packed(payload2, offsets2, ctx, init, out=(bbuf, sbuf))
    t2 = time.perf_counter()
    if it:
        best = [min(best[0], t1 - t0), min(best[1], t2 - t1)]
assert payload2.tobytes() == payload.tobytes() and (offsets2 == offsets).all()
assert (dbits == bits).all() and (status == 0).all()
te, td = best
print("... bins as bits on the host side (%.0f MB in): encode %.1f ms = %.2f Gbin/s, decode %.1f ms = %.2f Gbin/s, round trip %.2f Gbin/s"
      % (1.125 * n / 1e6, te * 1e3, n / te / 1e9, td * 1e3, n / td / 1e9, n / (te + td) / 1e9))
