"""N > 1 path on CPU: world size 2 over gloo.  Each rank 'codes' its shard with the oracle (standing in for the GPU
engine, which has no CPU path), then the real sharding / gather code runs and the root checks that the stitched
payload + offsets equal the single-process result."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, S, B, C, out_q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from isscabac_amd import sharding
    from oracle.pyoracle import Oracle
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    orc = Oracle()
    lo, hi = sharding.shard_range(S, rank, world)
    bins, ctx = orc.synth(0xCABAC003, lo, hi - lo, B, C)  # counter-based generator: shard-local generation
    init = np.zeros(C, np.uint8)
    payload, offsets = orc.encode_batch(bins, ctx, init)
    p_all, o_all = sharding.gather_payload(torch.from_numpy(payload), torch.from_numpy(offsets.astype(np.int64)), dist)
    if rank == 0:
        out_q.put((p_all.numpy().tobytes(), o_all.numpy().tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_ranges_tile_the_stream_range():
    from isscabac_amd import sharding
    for S in (1, 7, 64, 1000, 1 << 20):
        for world in (1, 2, 3, 4, 8):
            edges = [sharding.shard_range(S, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == S
            for a, b in zip(edges, edges[1:]):
                assert a[1] == b[0]
            sizes = [hi - lo for lo, hi in edges]
            assert max(sizes) - min(sizes) <= 1


def test_two_rank_gather_equals_single_process(oracle):
    import torch.multiprocessing as mp
    S, B, C = 37, 600, 64
    ctxm = mp.get_context("spawn")
    q = ctxm.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctxm.Process(target=_worker, args=(r, 2, port, S, B, C, q)) for r in range(2)]
    for p in procs:
        p.start()
    got_payload, got_offsets = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    bins, ctx = oracle.synth(0xCABAC003, 0, S, B, C)
    want_payload, want_offsets = oracle.encode_batch(bins, ctx, np.zeros(C, np.uint8))
    assert got_offsets == want_offsets.tolist()
    assert got_payload == want_payload.tobytes()
