#!/usr/bin/env python3
"""Per-bin handles used side by side: N handles decoding in turn (every call goes to another handle's live kernel) and a batch
call right behind a per-bin call -- does a spinning live kernel ever stand in the way of another kernel?
usage: python tools/live_interleave.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401
from isscabac_amd import Engine, Stream

eng = Engine(0)
rng = np.random.default_rng(3)
n = 3000
ctx = rng.integers(0, 3, n).astype(np.uint8)
bins = rng.integers(0, 2, n).astype(np.uint8)
init = np.zeros(3, np.uint8)
enc = Stream(eng, init)
enc.encode_start()
enc.encode_bins(bins, ctx)
payload = enc.encode_finish()
for nh in (1, 2, 4, 8, 16):
    hs = [Stream(eng, init) for _ in range(nh)]
    for h in hs:
        h.decode_start(payload)
    t0 = time.perf_counter()
    worst = 0.0
    for i in range(n):
        for h in hs:
            t1 = time.perf_counter()
            b = h.decode_bins(ctx[i:i + 1])[0]
            worst = max(worst, time.perf_counter() - t1)
            assert b == bins[i]
    dt = time.perf_counter() - t0
    print("%2d handles decoding in turn: %.1f us per call, worst call %.0f us" % (nh, dt / (n * nh) * 1e6, worst * 1e6))
    for h in hs:
        assert h.decode_finish() == 0
        h.close()
# a batch call right behind a per-bin call (the handle's live kernel is still polling)
h = Stream(eng, init)
h.decode_start(payload)
B = 256
bb = rng.integers(0, 2, (64, B)).astype(np.uint8)
cc = rng.integers(0, 3, (64, B)).astype(np.uint8)
eng.encode(bb, cc, init)
lat = []
for i in range(200):
    h.decode_bins(ctx[i:i + 1])
    t1 = time.perf_counter()
    eng.encode(bb, cc, init)
    lat.append(time.perf_counter() - t1)
lat2 = []
for i in range(200):
    t1 = time.perf_counter()
    eng.encode(bb, cc, init)
    lat2.append(time.perf_counter() - t1)
print("batch encode of 64 x %d bins right behind a decodeBin: median %.0f us, worst %.0f us; without a live kernel: median %.0f us, worst %.0f us"
      % (B, np.median(lat) * 1e6, max(lat) * 1e6, np.median(lat2) * 1e6, max(lat2) * 1e6))
