#!/usr/bin/env python3
"""Where does the decode kernel wait?  Needs a library built with -DISSCABAC_EXP_CLK=1 (ISSCABAC_LIB=...): the kernel then
writes, per wave, the s_memtime ticks spent in the payload wait of every chunk boundary, in the ctx wait of every round and
in total into the status words of its first three streams."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from isscabac_amd import Engine
eng = Engine(0)
dev = torch.device("cuda:0")
S, B, C = 1 << 20, 4096, 64
bins = torch.empty((S, B), dtype=torch.uint8, device=dev); ctx = torch.empty((S, B), dtype=torch.uint8, device=dev)
cap = S * 1100
payload = torch.empty(cap, dtype=torch.uint8, device=dev); offsets = torch.empty(S + 1, dtype=torch.int64, device=dev)
out = torch.empty((S, B), dtype=torch.uint8, device=dev); status = torch.empty((S,), dtype=torch.int32, device=dev)
eng.set_profiling(True)
eng.synth_dev(0xCABAC003, 0, S, B, B, C, False, bins.data_ptr(), ctx.data_ptr())
init = np.zeros(C, np.uint8)
for _ in range(2):
    total = eng.encode_dev(bins.data_ptr(), ctx.data_ptr(), S, B, B, init, payload.data_ptr(), cap, offsets.data_ptr())
    eng.decode_dev(payload.data_ptr(), total, offsets.data_ptr(), ctx.data_ptr(), S, B, B, init, out.data_ptr(), status.data_ptr())
    d = eng.last_kernel_ms()["decode"]
st = status.cpu().numpy().astype(np.int64).reshape(-1, 64)
aw, cx, tot, body = st[:, 0] & 0xFFFFFFFF, st[:, 1] & 0xFFFFFFFF, st[:, 2] & 0xFFFFFFFF, st[:, 3] & 0xFFFFFFFF
cst, cld = st[:, 4] & 0xFFFFFFFF, st[:, 5] & 0xFFFFFFFF
print("store passes: %.1f%% of a wave's life (%.0f ticks each), ctx load issue: %.1f%% (%.0f ticks each)" % (
    100 * cst.mean() / tot.mean(), cst.mean() / (B / 128), 100 * cld.mean() / tot.mean(), cld.mean() / (B / 64)))
print("plain chunk bodies: %.0f ticks per wave = %.1f%% of its life, %.1f ticks per bin" % (body.mean(), 100 * body.mean() / tot.mean(), body.mean() / B))
print("decode %.3f ms; per wave: total %.0f ticks (%.2f GHz if a wave lives kernel/4), payload waits %.1f%%, ctx waits %.1f%% (means over %d waves); p50/p90/p99 of payload-wait share: %s"
      % (d, tot.mean(), tot.mean() / (d / 4 * 1e6), 100 * aw.mean() / tot.mean(), 100 * cx.mean() / tot.mean(), len(tot),
         np.percentile(aw / tot, [50, 90, 99]).round(3)))
