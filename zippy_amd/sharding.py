"""Sharding of a batch of independent buffers over the GPUs of one node.

zippy's compress()/uncompress() are pure functions of one buffer (src/zippy.nim:11-16,
100-104): there is no exchange step inside the path, so a batch shards by contiguous
ranges of the buffer index and every rank runs the single-GPU engine on its own range
(SURVEY.md 8e).  The only communication is moving WHOLE buffers when a batch starts
or has to end on one rank: point-to-point sends of contiguous byte ranges (RCCL
send/recv over xGMI on GPUs -- root to each peer over its own link -- or gloo on CPU
in the tests), plus one all_gather of the per-buffer compressed lengths.

One process per GPU (torch.distributed); tensors are uint8, on the device the
process group works with.
"""
import torch
import torch.distributed as dist


def shard_range(n_buffers, rank, world):
    """Contiguous range [lo, hi) of buffer indices owned by `rank`."""
    base, extra = divmod(n_buffers, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def scatter_fixed(batch, n_buffers, buf_bytes, root=0, group=None, device=None):
    """Root holds `batch` (uint8 tensor [n_buffers * buf_bytes]); every rank gets its
    contiguous shard as a new tensor.  One send per peer, all in flight together."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    lo, hi = shard_range(n_buffers, rank, world)
    if rank == root:
        ops = []
        for r in range(world):
            rlo, rhi = shard_range(n_buffers, r, world)
            if r != root and rhi > rlo:
                ops.append(dist.P2POp(dist.isend, batch[rlo * buf_bytes:rhi * buf_bytes], r, group))
        mine = batch[lo * buf_bytes:hi * buf_bytes].clone()
        for w in (dist.batch_isend_irecv(ops) if ops else []):
            w.wait()
        return mine
    mine = torch.empty((hi - lo) * buf_bytes, dtype=torch.uint8, device=device)
    if hi > lo:
        for w in dist.batch_isend_irecv([dist.P2POp(dist.irecv, mine, root, group)]):
            w.wait()
    return mine


def gather_variable(local_bytes, local_lens, root=0, group=None):
    """Every rank holds its shard's outputs back to back in `local_bytes` (uint8) with
    per-buffer lengths `local_lens` (int64 tensor).  Returns on root (bytes, lens) for the
    whole batch in buffer order, elsewhere (None, lens).  The lengths travel in one
    all_gather; the payloads in one send per peer."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    counts = torch.tensor([local_lens.numel()], dtype=torch.int64, device=local_lens.device)
    all_counts = [torch.zeros_like(counts) for _ in range(world)]
    dist.all_gather(all_counts, counts, group=group)
    all_counts = [int(c.item()) for c in all_counts]
    width = max(all_counts) if all_counts else 0
    padded = torch.zeros(width, dtype=torch.int64, device=local_lens.device)
    padded[:local_lens.numel()] = local_lens
    gathered = [torch.zeros_like(padded) for _ in range(world)]
    dist.all_gather(gathered, padded, group=group)
    lens = torch.cat([g[:c] for g, c in zip(gathered, all_counts)])
    totals = [int(g[:c].sum().item()) for g, c in zip(gathered, all_counts)]
    if rank != root:
        if totals[rank]:
            for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, local_bytes[:totals[rank]], root, group)]):
                w.wait()
        return None, lens
    out = torch.empty(sum(totals), dtype=torch.uint8, device=local_bytes.device)
    ops, off = [], 0
    for r in range(world):
        if r == root:
            out[off:off + totals[r]] = local_bytes[:totals[r]]
        elif totals[r]:
            ops.append(dist.P2POp(dist.irecv, out[off:off + totals[r]], r, group))
        off += totals[r]
    for w in (dist.batch_isend_irecv(ops) if ops else []):
        w.wait()
    return out, lens


def gather_fixed(local, n_buffers, buf_bytes, root=0, group=None):
    """Inverse of scatter_fixed: every rank holds its shard (uint8 [(hi-lo) * buf_bytes]); root
    gets the whole batch in buffer order (elsewhere None).  One receive per peer, all in flight."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    lo, hi = shard_range(n_buffers, rank, world)
    if rank != root:
        if hi > lo:
            for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, local[:(hi - lo) * buf_bytes], root, group)]):
                w.wait()
        return None
    out = torch.empty(n_buffers * buf_bytes, dtype=torch.uint8, device=local.device)
    ops = []
    for r in range(world):
        rlo, rhi = shard_range(n_buffers, r, world)
        if r == root:
            out[rlo * buf_bytes:rhi * buf_bytes] = local[:(rhi - rlo) * buf_bytes]
        elif rhi > rlo:
            ops.append(dist.P2POp(dist.irecv, out[rlo * buf_bytes:rhi * buf_bytes], r, group))
    for w in (dist.batch_isend_irecv(ops) if ops else []):
        w.wait()
    return out


def scatter_variable(batch_bytes, lens, n_buffers, root=0, group=None, device=None):
    """Inverse of gather_variable: root holds the buffers of the whole batch back to back in
    `batch_bytes` (uint8) with `lens` (int64 tensor [n_buffers]); every rank gets
    (its shard's bytes back to back, its shard's lens).  The lengths travel in one broadcast, the
    payloads in one send per peer."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    if rank != root:
        lens = torch.zeros(n_buffers, dtype=torch.int64, device=device)
    dist.broadcast(lens, src=root, group=group)
    ends = torch.cumsum(lens, 0).cpu()
    def span(r):
        rlo, rhi = shard_range(n_buffers, r, world)
        b0 = int(ends[rlo - 1]) if rlo else 0
        b1 = int(ends[rhi - 1]) if rhi else 0
        return rlo, rhi, b0, b1
    lo, hi, b0, b1 = span(rank)
    if rank == root:
        ops = []
        for r in range(world):
            _, _, r0, r1 = span(r)
            if r != root and r1 > r0:
                ops.append(dist.P2POp(dist.isend, batch_bytes[r0:r1], r, group))
        mine = batch_bytes[b0:b1].clone()
        for w in (dist.batch_isend_irecv(ops) if ops else []):
            w.wait()
        return mine, lens[lo:hi]
    mine = torch.empty(b1 - b0, dtype=torch.uint8, device=device)
    if b1 > b0:
        for w in dist.batch_isend_irecv([dist.P2POp(dist.irecv, mine, root, group)]):
            w.wait()
    return mine, lens[lo:hi]


def _engine_stream(plan, device):
    """The stream the plan's engine launches on, as a torch stream (zh_stream; None for CPU tensors: the emulator's
    launches are synchronous).  It need not be torch's current stream -- an Engine made without one owns its own --, so
    the two functions below order the two explicitly instead of assuming they are the same."""
    if device.type != "cuda":
        return None
    ptr = plan.engine.stream()
    return torch.cuda.ExternalStream(ptr, device=device) if ptr else torch.cuda.default_stream(device)


def pack_plan(plan, slots, n_buffers):
    """The results of a device-resident plan (zippy_amd Plan; `slots`: the uint8 tensor its run wrote into) packed back
    to back on the device -- zh_plan_pack: two launches on the engine's stream, no per-buffer copy, the lengths are
    the ones the run left on the device -> (bytes, lens) as gather_variable takes them.  One host read: the total."""
    packed = torch.empty(max(1, slots.numel()), dtype=torch.uint8, device=slots.device)
    offs = torch.zeros(n_buffers + 1, dtype=torch.int64, device=slots.device)
    es = _engine_stream(plan, slots.device)
    if es is not None:
        cur = torch.cuda.current_stream(slots.device)
        es.wait_stream(cur)  # `offs` is zeroed, `slots` may have been produced, on torch's stream
        packed.record_stream(es)
        offs.record_stream(es)
    plan.pack(slots.data_ptr(), packed.data_ptr(), packed.numel(), offs.data_ptr())
    if es is not None:
        cur.wait_stream(es)  # what follows on torch's stream (the reads below, the sends) sees the packed bytes
        es.synchronize()
    total = int(offs[n_buffers].item())
    assert total <= packed.numel()
    return packed[:total], offs[1:] - offs[:-1]


def unpack_into_plan(plan, packed, lens, slots):
    """Inverse: streams back to back (as scatter_variable delivers them) into an uncompress plan's source slots, the
    lengths with them (zh_plan_unpack)."""
    offs = torch.zeros(lens.numel() + 1, dtype=torch.int64, device=slots.device)
    offs[1:] = torch.cumsum(lens.to(slots.device), 0)
    es = _engine_stream(plan, slots.device)
    if es is not None:
        es.wait_stream(torch.cuda.current_stream(slots.device))  # `offs` and `packed` were made on torch's stream
        offs.record_stream(es)
        packed.record_stream(es)
    plan.unpack(packed.data_ptr(), offs.data_ptr(), slots.data_ptr())
    return offs  # (kept alive by the caller until the plan has run)
