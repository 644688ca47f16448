"""cabacDemo.m on the batch engine (BASELINE config 1): N Gaussian AR(1) symbols, mid-tread quantised to Nq levels, TU
binarisation, three contexts chosen from the previous symbol's first bin (ref: CABAC/cabacDemo.m:8-200), one stream.

    python -m isscabac_amd.demo [N=10000] [Nq=4] [filename=test.bin]

MATLAB's rng(0,'twister') + randn stream is not reproduced (numpy's generator differs), so the bit count differs from a
MATLAB run of the demo; the coding itself is bit-exact for the same symbols (tests/test_gpu_parity.py).
"""
import sys

import numpy as np

from .engine import Engine, SymbolParams, init_by_prob


def cabacDemo(N=10000, Nq=4, filename="test.bin", fancyInit=False, seed=0, engine=None):
    eng = engine or Engine(0)
    rng = np.random.default_rng(seed)
    symbols = np.abs(rng.normal(size=N))
    symbols[1:] += 0.8 * symbols[:-1]  # cabacDemo.m:29-30
    delta = np.quantile(symbols, 0.99) / Nq
    symbols = np.minimum(np.floor(symbols / delta + 0.5), Nq - 1).astype(np.uint8)  # mid-tread quantiser, :35-37
    if not fancyInit:
        ctx_init = np.full(3, 0.5)
    else:  # :51-70: p(0) over all bins, and of the first bin given the previous symbol's first bin
        first = (symbols > 0).astype(np.uint8)  # first TU bin is 1 unless the symbol is 0
        nbins_each = np.where(symbols == Nq - 1, Nq - 1, symbols + 1)
        zeros_total = int((symbols < Nq - 1).sum())
        ctx_init = np.array([zeros_total / nbins_each.sum(),
                             (first[1:][first[:-1] == 1] == 0).mean() if (first[:-1] == 1).any() else 0.5,
                             (first[1:][first[:-1] == 0] == 0).mean() if (first[:-1] == 0).any() else 0.5])
    states = init_by_prob(ctx_init)
    params = SymbolParams.demo(Nq)
    payload, offsets, nbins = eng.encode_symbols(symbols[None, :], params, states)
    with open(filename, "wb") as f:
        f.write(payload.tobytes())
    decoded, status = eng.decode_symbols(payload, offsets, N, params, states)
    assert status[0] == 0 and (decoded[0, :N] == symbols).all(), "Mismatch"
    nbits = 8 * payload.size
    print("CABAC needs %d bits for encoding the test sequence (%d symbols, %d bins)." % (nbits, N, int(nbins[0])))
    return nbits


if __name__ == "__main__":
    a = sys.argv[1:]
    cabacDemo(int(a[0]) if a else 10000, int(a[1]) if len(a) > 1 else 4, a[2] if len(a) > 2 else "test.bin")
