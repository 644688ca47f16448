import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from isscabac_amd import Engine
eng = Engine(0)
rng = np.random.default_rng(1)
def run(name, bins, ctx, init):
    out = {}
    for t in ("throughput", "latency"):
        eng.set_tuning(t)
        payload, offsets = eng.encode(bins, ctx, init)
        out[t] = (payload, offsets)
    a, b = out["throughput"], out["latency"]
    same = a[0].tobytes() == b[0].tobytes() and (a[1] == b[1]).all()
    print(name, "OK" if same else "DIFF", "lens", np.diff(a[1])[:4], np.diff(b[1])[:4])
    if not same:
        for s in range(min(2, bins.shape[0])):
            pa = a[0][a[1][s]:a[1][s+1]]; pb = b[0][b[1][s]:b[1][s+1]]
            print("  stream", s, pa[:12].tobytes().hex(), "|", pb[:12].tobytes().hex())
S = 64
for B in (16, 32, 64):
    init = rng.integers(0, 126, 64).astype(np.uint8)
    bins = rng.integers(0, 2, (S, B)).astype(np.uint8)
    run("B=%d distinct ctx" % B, bins, np.tile(np.arange(B, dtype=np.uint8) % 64, (S, 1)), init)
    run("B=%d one ctx" % B, bins, np.zeros((S, B), np.uint8), init)
    run("B=%d two ctx alternate" % B, bins, np.tile((np.arange(B) & 1).astype(np.uint8), (S, 1)), init)
    run("B=%d random ctx" % B, bins, rng.integers(0, 64, (S, B)).astype(np.uint8), init)
    run("B=%d zero bins one ctx" % B, np.zeros((S, B), np.uint8), np.zeros((S, B), np.uint8), init)
