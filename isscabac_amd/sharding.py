"""Stream sharding across the GPUs of one node and the final bitstream gather.

Streams are independent units (one reference handle = one stream, ref: CABAC/SimpleCABACMex.cpp:69-80), so rank r of
N codes the contiguous block [lo, hi) of the global stream range and nothing is exchanged while coding.  The only
communication is the gather of the per-rank payloads to the root at the end: sizes first (all_gather of one int64
per rank), then every rank sends its payload straight to the root, which posts all receives at once -- on MI355X
that keeps all seven xGMI links of the root busy concurrently instead of serialising a ring.  Every receive lands in
its slice of one buffer the root allocates once; offsets are rebased on the sending device.  Works on any
torch.distributed backend ("nccl" = RCCL on the GPU box, "gloo" in the CPU tests).
"""

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


def gather_payload(payload, offsets, dist, root=0, out=None):
    """payload: 1-D uint8 torch tensor holding this rank's bitstreams, offsets: int64 tensor (S_r + 1).
    Returns (payload_all, offsets_all) on the root (global stream order), (None, None) elsewhere.

    The sizes are known after one all_gather, so the root allocates the job's payload ONCE and posts a receive for every
    peer straight into that peer's slice of it (no second copy of the bitstream, no torch.cat); every rank rebases its
    own offsets on its device before sending (base = bytes of the ranks before it), so the root receives final values
    into the slices of one offsets array and no offset ever visits the host.  `out` = (payload_all, offsets_all)
    buffers of at least the job's size on the root, if the caller keeps them across gathers."""
    import torch
    rank, world = dist.get_rank(), dist.get_world_size()
    dev = payload.device
    n_local = offsets.numel() - 1
    mine = torch.tensor([int(payload.numel()), n_local], dtype=torch.int64, device=dev)
    sizes = torch.zeros((world, 2), dtype=torch.int64, device=dev)
    if dist.get_backend() == "nccl":  # RCCL: one collective into the table; gloo has no all_gather_into_tensor
        dist.all_gather_into_tensor(sizes, mine)
    else:
        dist.all_gather([sizes[r] for r in range(world)], mine)
    sizes = sizes.cpu().tolist()  # 16 bytes per rank: the only host visit
    byte_base = [0]
    strm_base = [0]
    for nb, ns in sizes:
        byte_base.append(byte_base[-1] + nb)
        strm_base.append(strm_base[-1] + ns)
    if rank != root:
        ops = []
        if payload.numel():
            ops.append(dist.P2POp(dist.isend, payload, root))
        if n_local:
            ops.append(dist.P2POp(dist.isend, offsets[1:] + byte_base[rank], root))
        for q in (dist.batch_isend_irecv(ops) if ops else []):
            q.wait()
        return None, None
    if out is not None:
        p_all, o_all = out[0][:byte_base[-1]], out[1][:strm_base[-1] + 1]
    else:
        p_all = torch.empty(byte_base[-1], dtype=torch.uint8, device=dev)
        o_all = torch.empty(strm_base[-1] + 1, dtype=torch.int64, device=dev)
    ops = []
    for r in range(world):
        if r == root:
            continue
        if sizes[r][0]:
            ops.append(dist.P2POp(dist.irecv, p_all[byte_base[r]:byte_base[r + 1]], r))
        if sizes[r][1]:
            ops.append(dist.P2POp(dist.irecv, o_all[strm_base[r] + 1:strm_base[r + 1] + 1], r))
    # one batched group: with RCCL this is a single ncclGroupStart/End, all peers' links in flight together
    reqs = dist.batch_isend_irecv(ops) if ops else []
    # the root's own shard moves while the receives are in flight
    o_all[:1] = 0
    p_all[byte_base[root]:byte_base[root + 1]].copy_(payload)
    torch.add(offsets[1:], byte_base[root], out=o_all[strm_base[root] + 1:strm_base[root + 1] + 1])
    for q in reqs:
        q.wait()
    return p_all, o_all
