"""N > 1 path on CPU: world size 2 over gloo.  Each rank 'codes' its shard with the oracle (standing in for the GPU
engine, which has no CPU path), then the real sharding / gather code runs and the root checks that the stitched
payload + offsets equal the single-process result."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker_plan(rank, world, port, S, per_gpu, B, C, scaling, chunk, out_q):
    """what bench.py does per rank, with the oracle standing in for the engine: plan -> launches -> concat -> gather"""
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from isscabac_amd import sharding
    from oracle.pyoracle import Oracle
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    orc = Oracle()
    init = np.zeros(C, np.uint8)
    parts = []
    for lo, hi in sharding.plan(S, per_gpu, rank, world, scaling, chunk):
        bins, ctx = orc.synth(0xCABAC005, lo, hi - lo, B, C, True)
        payload, offsets = orc.encode_batch(bins, ctx, init)
        parts.append((torch.from_numpy(payload), torch.from_numpy(offsets.astype(np.int64))))
    p_mine, o_mine = sharding.concat_launches(parts)
    p_all, o_all = sharding.gather_payload(p_mine, o_mine, dist)
    if rank == 0:
        out_q.put((p_all.numpy().tobytes(), o_all.numpy().tolist()))
    dist.barrier()
    dist.destroy_process_group()


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


def test_plan_strong_and_weak():
    from isscabac_amd import sharding
    for S, per_gpu, chunk in ((1 << 20, 1 << 20, 1 << 20), (8 << 20, 1 << 20, 1 << 20), (1000, 300, 128), (7, 3, 2)):
        for world in (1, 2, 4, 8):
            # strong: the launches of all ranks tile [0, S) exactly once, in rank order, none above `chunk`
            edges = [c for r in range(world) for c in sharding.plan(S, per_gpu, r, world, "strong", chunk)]
            assert edges[0][0] == 0 and edges[-1][1] == S
            assert all(a[1] == b[0] for a, b in zip(edges, edges[1:])) and all(0 < hi - lo <= chunk for lo, hi in edges)
            assert sharding.job_streams(S, per_gpu, world, "strong") == S
            # weak: every rank codes per_gpu streams, rank r's block starts at r * per_gpu
            for r in range(world):
                pl = sharding.plan(S, per_gpu, r, world, "weak", chunk)
                assert pl[0][0] == r * per_gpu and pl[-1][1] == (r + 1) * per_gpu
                assert sum(hi - lo for lo, hi in pl) == per_gpu
            assert sharding.job_streams(S, per_gpu, world, "weak") == per_gpu * world
    # BASELINE.json's shapes: config 3 over 8 GPUs = 128 Ki streams each; config 5 = 1 Mi each, one launch
    assert sharding.plan(1 << 20, 1 << 20, 3, 8, "strong", 1 << 20) == [(3 << 17, 4 << 17)]
    assert sharding.plan(8 << 20, 1 << 20, 7, 8, "strong", 1 << 20) == [(7 << 20, 8 << 20)]
    assert len(sharding.plan(8 << 20, 1 << 20, 0, 1, "strong", 1 << 20)) == 8
    with pytest.raises(ValueError):
        sharding.plan(8, 8, 0, 1, "both", 8)


@pytest.mark.parametrize("scaling", ["strong", "weak"])
def test_two_rank_planned_job_equals_single_process(oracle, scaling):
    """strong-mode (and weak-mode) shard arithmetic end to end: two ranks, several launches each, stitched and gathered;
    the root's payload + offsets must equal one process coding the whole job"""
    import torch.multiprocessing as mp
    S, per_gpu, B, C, chunk = 45, 19, 320, 64, 8
    ctxm = mp.get_context("spawn")
    q = ctxm.Queue()
    port = 31500 + os.getpid() % 2000 + (7 if scaling == "weak" else 0)
    procs = [ctxm.Process(target=_worker_plan, args=(r, 2, port, S, per_gpu, B, C, scaling, chunk, q)) for r in range(2)]
    for p in procs:
        p.start()
    got_payload, got_offsets = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    from isscabac_amd import sharding
    n = sharding.job_streams(S, per_gpu, 2, scaling)
    bins, ctx = oracle.synth(0xCABAC005, 0, n, B, C, True)
    want_payload, want_offsets = oracle.encode_batch(bins, ctx, np.zeros(C, np.uint8))
    assert got_offsets == want_offsets.tolist()
    assert got_payload == want_payload.tobytes()
