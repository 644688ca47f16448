import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from isscabac_amd import Engine
eng = Engine(0)
dev = torch.device("cuda:0")
S, B, C = 1 << 18, 4096, 64
for C in (64, 255):
    bins = torch.empty((S, B), dtype=torch.uint8, device=dev); ctx = torch.empty((S, B), dtype=torch.uint8, device=dev)
    eng.synth_dev(0xCABAC003, 0, S, B, B, 64, False, bins.data_ptr(), ctx.data_ptr())
    if C == 255:
        ctx.mul_(3)  # contexts 0..189 (state region of the > 64 contexts kernels)
    init = np.zeros(C, np.uint8)
    cap = S * 1100
    payload = torch.empty(cap, dtype=torch.uint8, device=dev); offsets = torch.empty(S + 1, dtype=torch.int64, device=dev)
    out = torch.empty((S, B), dtype=torch.uint8, device=dev); status = torch.empty((S,), dtype=torch.int32, device=dev)
    eng.set_profiling(True)
    for _ in range(2):
        total = eng.encode_dev(bins.data_ptr(), ctx.data_ptr(), S, B, B, init, payload.data_ptr(), cap, offsets.data_ptr())
        e = eng.last_kernel_ms()["encode"]
        eng.decode_dev(payload.data_ptr(), total, offsets.data_ptr(), ctx.data_ptr(), S, B, B, init, out.data_ptr(), status.data_ptr())
        d = eng.last_kernel_ms()["decode"]
    assert torch.equal(out, bins) and int(status.abs().sum()) == 0
    print("C=%d: encode %.3f ms decode %.3f ms -> %.1f Gbin/s round trip (kernels only)" % (C, e, d, S * B / (e + d) / 1e6))
