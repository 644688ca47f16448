#!/usr/bin/env python3
"""Rate of the PER-BIN command path -- what the reference's own callers do (ref: CABAC/cabacDemo.m:92-131,
ISS/+coder/cabacEncode.m:45-70: one encodeBin / decodeBin MEX call per bin, getNumBits now and then) -- through the
device-resident handle (isscabac_stream_*, ctypes) and through the MEX gateway compiled against the stub mex.h
(matlab/SimpleCABACMex.cpp, tests/mex_gateway.py).  BASELINE config 1's shape: one stream, TU-binarised random symbols,
3 contexts.  The reference's core loop does ~69 Mbin/s on one host core (bench.py cpu_baseline); a per-bin call into a GPU
engine is one launch (encode: queued, flushed lazily) or one launch + one synchronisation (decodeBin, getNumBits) per call --
this path is plumbing, the batch calls are the product.  Numbers go to profiles/ beside the batch figures."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch  # noqa: F401
from isscabac_amd import Engine, Stream

rng = np.random.default_rng(1)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
ctx = rng.integers(0, 3, n).astype(np.uint8)
bins = (rng.random(n) < np.array([0.2, 0.5, 0.8])[ctx]).astype(np.uint8)
init = np.zeros(3, np.uint8)
eng = Engine(0)

st = Stream(eng, init)
st.encode_start()
t0 = time.perf_counter()
for i in range(n):
    st.encode_bins(bins[i:i + 1], ctx[i:i + 1])
t1 = time.perf_counter()
for i in range(2000):
    st.encode_bins(bins[i:i + 1], ctx[i:i + 1])
    st.num_bits()
t2 = time.perf_counter()
payload = st.encode_finish()
st2 = Stream(eng, init)
st2.decode_start(payload)
t3 = time.perf_counter()
got = np.array([st2.decode_bins(ctx[i:i + 1])[0] for i in range(n)], np.uint8)
t4 = time.perf_counter()
assert (got == bins).all()
print("isscabac_stream_* per-bin calls, %d bins: encodeBin %.1f kbin/s (%.1f us per call, queued), encodeBin + getNumBits %.1f kbin/s "
      "(%.1f us per pair), decodeBin %.1f kbin/s (%.1f us per call)"
      % (n, n / (t1 - t0) / 1e3, (t1 - t0) / n * 1e6, 2000 / (t2 - t1) / 1e3, (t2 - t1) / 2000 * 1e6, n / (t4 - t3) / 1e3, (t4 - t3) / n * 1e6))

try:
    from mex_gateway import Gateway
    gw = Gateway()
    import tempfile
    fn = os.path.join(tempfile.mkdtemp(), "perbin.bin")
    m = min(n, 10000)
    (h,) = gw.call("initByProb", fn, np.array([0.5, 0.5, 0.5]), nlhs=1)
    gw.call("encodeStart", h)
    t0 = time.perf_counter()
    for i in range(m):
        gw.call("encodeBin", h, float(bins[i]), float(ctx[i]))
    t1 = time.perf_counter()
    gw.call("encodeFinish", h)
    (h2,) = gw.call("initByProb", fn, np.array([0.5, 0.5, 0.5]), nlhs=1)
    gw.call("decodeStart", h2)
    t2 = time.perf_counter()
    got = [int(gw.call("decodeBin", h2, float(ctx[i]), nlhs=1)[0][0, 0]) for i in range(m)]
    t3 = time.perf_counter()
    gw.call("decodeFinish", h2)
    assert got == bins[:m].tolist()
    print("MEX gateway (stub mex.h, python driver) per-bin commands, %d bins: encodeBin %.1f kbin/s (%.1f us per call), decodeBin %.1f kbin/s "
          "(%.1f us per call)" % (m, m / (t1 - t0) / 1e3, (t1 - t0) / m * 1e6, m / (t3 - t2) / 1e3, (t3 - t2) / m * 1e6))
except Exception as exc:  # the gateway needs g++ on the box
    print("MEX gateway leg skipped:", repr(exc))
