"""Stream sharding across the GPUs of one node and the final bitstream gather.

Streams are independent units (one reference handle = one stream, ref: CABAC/SimpleCABACMex.cpp:69-80), so rank r of
N codes the contiguous block [lo, hi) of the global stream range and nothing is exchanged while coding.  The only
communication is the gather of the per-rank payloads to the root at the end: sizes first (all_gather of one int64
per rank), then every rank sends its payload straight to the root, which posts all receives at once -- on MI355X
that keeps all seven xGMI links of the root busy concurrently instead of serialising a ring.  Works on any
torch.distributed backend ("nccl" = RCCL on the GPU box, "gloo" in the CPU tests).
"""
import numpy as np


def shard_range(n_streams, rank, world):
    """Contiguous block of streams for `rank`; blocks differ by at most one stream and concatenate in rank order."""
    base, rem = divmod(int(n_streams), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def chunk_ranges(lo, hi, chunk):
    """[lo, hi) cut into consecutive launches of at most `chunk` streams (what fits one GPU's HBM at a time)."""
    lo, hi, chunk = int(lo), int(hi), int(chunk)
    assert chunk > 0 and hi >= lo
    return [(a, min(a + chunk, hi)) for a in range(lo, hi, chunk)]


def plan(global_streams, per_gpu_streams, rank, world, scaling, chunk):
    """Which global streams `rank` of `world` codes, as a list of (first, last + 1) launches.

    scaling == "strong": the workload is `global_streams` streams whatever the number of GPUs; rank r takes the
    contiguous block shard_range(global_streams, r, world) (global order = concatenation of the shards).
    scaling == "weak": every GPU codes `per_gpu_streams` streams; rank r's block starts at r * per_gpu_streams (the
    job grows with the number of GPUs).  Either way the block is run `chunk` streams at a time."""
    if scaling == "strong":
        lo, hi = shard_range(global_streams, rank, world)
    elif scaling == "weak":
        lo, hi = rank * int(per_gpu_streams), (rank + 1) * int(per_gpu_streams)
    else:
        raise ValueError("scaling must be 'strong' or 'weak', not %r" % (scaling,))
    return chunk_ranges(lo, hi, chunk)


def job_streams(global_streams, per_gpu_streams, world, scaling):
    """streams the whole job codes per step"""
    return int(global_streams) if scaling == "strong" else int(per_gpu_streams) * int(world)


def global_offsets(shard_offsets, shard_sizes):
    """Stitch per-rank offsets (each starting at 0, S_r + 1 entries) into the global S + 1 offsets array."""
    out = [np.zeros(1, np.uint64)]
    base = 0
    for off, size in zip(shard_offsets, shard_sizes):
        off = np.asarray(off, np.uint64)
        assert int(off[-1]) == int(size)
        out.append(off[1:] + np.uint64(base))
        base += int(size)
    return np.concatenate(out)


def concat_launches(parts):
    """One rank's consecutive launches [(payload, offsets), ...] (torch tensors: uint8 payload of exactly offsets[-1]
    bytes, int64 offsets with S_i + 1 entries starting at 0) as one (payload, offsets) pair in stream order."""
    import torch
    if len(parts) == 1:
        return parts[0]
    base, offs = 0, [torch.zeros(1, dtype=torch.int64, device=parts[0][1].device)]
    for payload, offsets in parts:
        assert int(offsets[-1]) == payload.numel()
        offs.append(offsets[1:] + base)
        base += payload.numel()
    return torch.cat([p for p, _ in parts]), torch.cat(offs)


def gather_payload(payload, offsets, dist, root=0):
    """payload: 1-D uint8 torch tensor holding this rank's bitstreams, offsets: int64 tensor (S_r + 1).
    Returns (payload_all, offsets_all) on the root (global stream order), (None, None) elsewhere."""
    import torch
    rank, world = dist.get_rank(), dist.get_world_size()
    dev = payload.device
    n_local = offsets.numel() - 1
    mine = torch.tensor([int(payload.numel()), n_local], dtype=torch.int64, device=dev)
    sizes = [torch.zeros(2, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(sizes, mine)
    sizes = [s.cpu().tolist() for s in sizes]
    if rank == root:
        parts, offs, ops = [], [], []
        for r in range(world):
            if r == root:
                parts.append(payload)
                offs.append(offsets)
                continue
            p = torch.empty(sizes[r][0], dtype=torch.uint8, device=dev)
            o = torch.empty(sizes[r][1] + 1, dtype=torch.int64, device=dev)
            ops.append(dist.P2POp(dist.irecv, p, r))
            ops.append(dist.P2POp(dist.irecv, o, r))
            parts.append(p)
            offs.append(o)
        # one batched group: with RCCL this is a single ncclGroupStart/End, all peers' links in flight together
        for q in (dist.batch_isend_irecv(ops) if ops else []):
            q.wait()
        all_off = global_offsets([o.cpu().numpy() for o in offs], [s[0] for s in sizes])
        return torch.cat(parts), torch.from_numpy(all_off.astype(np.int64)).to(dev)
    ops = [dist.P2POp(dist.isend, payload, root), dist.P2POp(dist.isend, offsets, root)]
    for q in dist.batch_isend_irecv(ops):
        q.wait()
    return None, None
