#!/usr/bin/env python3
"""The <= 64 context kernels of config 3 at reduced occupancy: a timing-only build (tools/ab_patches/lds_pad.patch, ISSCABAC_LIB) pads
the workgroup's dynamic LDS so that fewer four-wave workgroups fit a CU -- what the instances above 64 contexts would run at if
occupancy were all that separates them from the 64-context kernel.   ISSCABAC_LIB=tools/ab/variant_ldspad.so python tools/occupancy_curve.py"""
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
eng.synth_dev(0xCABAC003, 0, S, B, B, 64, False, bins.data_ptr(), ctx.data_ptr())
init = np.zeros(C, np.uint8)
# granules of 1280 bytes, 128 per CU: 32 -> 4 workgroups (16 waves), 42 -> 3 (12), 64 -> 2 (8), 128 -> 1 (4)
for gran, waves in ((0, 16), (42, 12), (64, 8), (128, 4)):
    os.environ["ISSCABAC_EXP_LDS_PAD"] = str(gran * 1280)
    for _ in range(2):
        total = eng.encode_dev(bins.data_ptr(), ctx.data_ptr(), S, B, B, init, payload.data_ptr(), cap, offsets.data_ptr())
        e = eng.last_kernel_ms()["encode"]
        eng.decode_dev(payload.data_ptr(), total, offsets.data_ptr(), ctx.data_ptr(), S, B, B, init, out.data_ptr(), status.data_ptr())
        d = eng.last_kernel_ms()["decode"]
    assert torch.equal(out, bins) and int(status.abs().sum()) == 0
    print("%2d waves per CU (workgroup LDS padded to %3d granules): encode %.3f ms decode %.3f ms -> %.1f Gbin/s round trip (kernels only)" % (
        waves, gran, e, d, S * B / (e + d) / 1e6))
