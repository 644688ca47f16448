#!/usr/bin/env python3
"""Is it their code or their data that separates the <= 128 context instances from the 64-context kernels at equal occupancy?
Config 3's 64-context batch through (a) its own kernels, (b) the same held to 12 waves per CU by LDS padding, (c) the <= 128
context instances (12 waves per CU by themselves) -- a timing-only build: tools/ab_patches/lds_pad.patch + force_cm3.patch.
    ISSCABAC_LIB=tools/ab/variant_cm3exp.so python tools/cm3_on_64.py"""
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
for name, env in (("64-context kernels, 16 waves per CU", {}), ("64-context kernels held to 12 waves", {"ISSCABAC_EXP_LDS_PAD": str(42 * 1280)}),
                  ("<= 128 context instances on the same batch (12 waves)", {"ISSCABAC_EXP_FORCE_CM3": "1"})):
    for k in ("ISSCABAC_EXP_LDS_PAD", "ISSCABAC_EXP_FORCE_CM3"):
        os.environ.pop(k, None)
    os.environ.update(env)
    for _ in range(3):
        total = eng.encode_dev(bins.data_ptr(), ctx.data_ptr(), S, B, B, init, payload.data_ptr(), cap, offsets.data_ptr())
        e = eng.last_kernel_ms()["encode"]
        ie = eng.last_instance("encode")
        eng.decode_dev(payload.data_ptr(), total, offsets.data_ptr(), ctx.data_ptr(), S, B, B, init, out.data_ptr(), status.data_ptr())
        d = eng.last_kernel_ms()["decode"]
        idd = eng.last_instance("decode")
    assert torch.equal(out, bins) and int(status.abs().sum()) == 0
    print("%-56s %s %.3f ms, %s %.3f ms -> %.1f Gbin/s" % (name + ":", ie, e, idd, d, S * B / (e + d) / 1e6))
