#!/usr/bin/env python3
"""One-off scale check: a batch whose payload exceeds 4 GiB (64-bit offsets everywhere).  5 Mi streams x 8 Ki bins,
50% bypass -> ~4.7 GB of payload; device round trip + sampled streams against the oracle.  Needs ~170 GB of HBM."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from isscabac_amd import Engine
from oracle.pyoracle import Oracle

S, B, C, seed = 5 << 20, 8192, 64, 0xCABAC005
dev = torch.device("cuda:0")
eng, orc = Engine(0), Oracle()
bins = torch.empty((S, B), dtype=torch.uint8, device=dev)
ctx = torch.empty((S, B), dtype=torch.uint8, device=dev)
eng.synth_dev(seed, 0, S, B, B, C, True, bins.data_ptr(), ctx.data_ptr())
init = np.zeros(C, np.uint8)
cap = S * 1100
payload = torch.empty(cap, dtype=torch.uint8, device=dev)
offsets = torch.empty(S + 1, dtype=torch.int64, device=dev)
t0 = time.perf_counter()
total = eng.encode_dev(bins.data_ptr(), ctx.data_ptr(), S, B, B, init, payload.data_ptr(), cap, offsets.data_ptr())
t1 = time.perf_counter()
print("payload bytes", total, "> 4 GiB:", total > (1 << 32), "encode %.1f ms" % ((t1 - t0) * 1e3))
assert total > (1 << 32)
out = torch.empty((S, B), dtype=torch.uint8, device=dev)
status = torch.empty((S,), dtype=torch.int32, device=dev)
eng.decode_dev(payload.data_ptr(), total, offsets.data_ptr(), ctx.data_ptr(), S, B, B, init, out.data_ptr(), status.data_ptr())
print("decode %.1f ms" % ((time.perf_counter() - t1) * 1e3))
assert int(status.abs().sum()) == 0 and torch.equal(out, bins)
off = offsets.cpu().numpy().astype(np.uint64)
for s in [0, 1, S // 2, S - 1] + list(range(4_600_000, 4_600_000 + 64)):  # streams that start beyond 4 GiB
    hb, hc = orc.synth(seed, s, 1, B, C, True)
    want = orc.encode_stream(hb[0], hc[0], init)
    got = payload[int(off[s]):int(off[s + 1])].cpu().numpy()
    assert got.tobytes() == want.tobytes(), s
print("ok: offsets up to", int(off[-1]), "sampled streams bit-exact, first checked offset beyond 4 GiB:", int(off[4_600_000]))
