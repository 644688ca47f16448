"""isscabac_multi: several devices behind one handle in one process.  The GPU box has one GPU, so
the engines of a handle share device 0 -- the shard arithmetic, the per-device host threads and the peer-copy gather are
the code a real multi-GPU node runs."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("ndev", [1, 2, 3])
def test_host_batches_equal_a_single_engine(oracle, ndev):
    from isscabac_amd import MultiEngine
    m = MultiEngine([0] * ndev)
    rng = np.random.default_rng(ndev)
    for S, B, C in ((1, 64, 3), (7, 200, 5), (257, 1040, 64)):
        bins, ctx = oracle.synth(0xCABAC005, 5, S, B, C, True)
        init = rng.integers(0, 126, C).astype(np.uint8)
        nb = rng.integers(0, B + 1, S).astype(np.uint32)
        edges = [m.shard(S, g) for g in range(ndev)]
        assert edges[0][0] == 0 and edges[-1][1] == S and all(a[1] == b[0] for a, b in zip(edges, edges[1:]))
        for nbins in (None, nb):
            want_p, want_o = oracle.encode_batch(bins, ctx, init, nbins)
            got_p, got_o = m.encode(bins, ctx, init, nbins)
            assert (got_o == want_o).all() and got_p.tobytes() == want_p.tobytes()
            dec, status = m.decode(want_p, want_o, ctx, init, nbins)
            mask = np.ones((S, B), bool) if nbins is None else np.arange(B)[None, :] < nbins[:, None]
            assert (status == 0).all() and (dec * mask == bins * mask).all()
    m.close()


@pytest.mark.parametrize("ndev", [2, 4])
def test_device_resident_shards_gathered_on_the_root(oracle, ndev):
    import torch
    from isscabac_amd import MultiEngine
    dev = torch.device("cuda:0")
    m = MultiEngine([0] * ndev)
    S, B, C = 1003, 512, 64  # not divisible by the number of devices
    bins, ctx = oracle.synth(0xCABAC003, 0, S, B, C)
    init = np.zeros(C, np.uint8)
    want_p, want_o = oracle.encode_batch(bins, ctx, init)
    tb, tc = [], []
    for g in range(ndev):
        lo, hi = m.shard(S, g)
        tb.append(torch.from_numpy(bins[lo:hi].copy()).to(dev))
        tc.append(torch.from_numpy(ctx[lo:hi].copy()).to(dev))
    cap = S * (B // 4 + 64)
    payload = torch.zeros(cap, dtype=torch.uint8, device=dev)
    offsets = torch.zeros(S + 1, dtype=torch.int64, device=dev)
    for _ in range(2):  # twice: the staging buffers are reused
        total = m.encode_dev([t.data_ptr() for t in tb], [t.data_ptr() for t in tc], S, B, B, init, payload.data_ptr(), cap,
                             offsets.data_ptr())
        torch.cuda.synchronize()
        assert total == want_p.size
        assert offsets.cpu().numpy().astype(np.uint64).tolist() == want_o.tolist()
        assert payload[:total].cpu().numpy().tobytes() == want_p.tobytes()
        payload.zero_()
        offsets.zero_()
    from isscabac_amd import EngineError
    with pytest.raises(EngineError):
        m.encode_dev([t.data_ptr() for t in tb], [t.data_ptr() for t in tc], S, B, B, init, payload.data_ptr(), 100,
                     offsets.data_ptr())
    m.close()
