"""Read sharding across the GPUs of one node (one process per GPU, torch.distributed).

The path has no exchange step: reads are independent units, the index is replicated in every GPU's HBM, and the only
cross-rank dependency is output order (the reference drains batches through a heap keyed by batch id,
minialign.c:4633-4645).  Hence: contiguous shards in input order, no collective on the data path, and a merge that simply
concatenates the per-rank SAM bodies in rank order.  The collectives used are the barrier / MAX of the timing in bench.py.

One caveat keeps this bit-exact with a single-process run: the reference carries one integer of state from read to read
(the length of the last loaded reference, read before it is refreshed -- minialign.c:3864).  A shard therefore needs the
value its predecessor ends with; `carry_chain` resolves it with one tiny all_gather of the per-rank (has_value, value)
pair after mapping, and tells which ranks must re-run their first reads (rare)."""

def shard_bounds(n_reads, rank, world):
    """contiguous [lo, hi) of the reads of `rank`; sizes differ by at most one"""
    base, rem = divmod(n_reads, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)

def merge_in_order(bodies):
    """bodies[r]: SAM text (no header) of rank r's contiguous shard -> whole-file body in input order"""
    return b''.join(bodies)

def carry_chain(dist, has_value, value, initial=0):
    """all ranks call this with the reference-length state their shard *ends* with (has_value = 0 if the shard never
    loaded a reference).  Returns the state each rank must *start* from."""
    import torch
    t = torch.tensor([int(has_value), int(value)], dtype=torch.int64)
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        out = [torch.zeros(2, dtype=torch.int64) for _ in range(dist.get_world_size())]
        dist.all_gather(out, t)
        rank = dist.get_rank()
    else:
        out = [t]; rank = 0
    cur = initial
    starts = []
    for r in range(len(out)):
        starts.append(cur)
        if int(out[r][0]):
            cur = int(out[r][1])
    return starts[rank]

def reduce_timing(dist, seconds, units):
    """(max over ranks of the elapsed time, sum over ranks of the processed units)"""
    import torch
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return seconds, units
    t = torch.tensor([seconds], dtype=torch.float64); dist.all_reduce(t, op=dist.ReduceOp.MAX)
    u = torch.tensor([float(units)], dtype=torch.float64); dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return float(t.item()), float(u.item())
