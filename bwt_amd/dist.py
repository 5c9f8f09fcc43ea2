"""Several GPUs, one process each (torch.distributed; backend "nccl" is RCCL).

Search: the index is replicated, reads are split into contiguous slices, no
exchange on the data path; results are gathered in read order only when the
caller asks for them on one rank.

Suffix-array build: the text is replicated (a quarter byte per base), every
rank sorts the suffixes of its own range of leading four-base strings -- a
contiguous range of suffix-array rows -- and the shards are then all-gathered
into place, so that every rank ends with the whole array and builds its own
copy of the index.  Ids are never exchanged before the sort:
each rank reads its suffixes straight off its copy of the text, which costs
less than moving four bytes per suffix over xGMI.

Nothing here touches the GPU directly: the local work is done by callables the
caller passes in (bwt_amd.gpu on a GPU box; the tests drive the same code on
CPU tensors over gloo).
"""
import numpy as np
import torch
import torch.distributed as dist


def _host_staged(group):
    """gloo moves host memory; device tensors are staged through the host for it
    (RCCL takes them as they are)."""
    return dist.get_backend(group) == "gloo"


def _broadcast(t, src, group):
    if t.is_cuda and _host_staged(group):
        h = t.cpu()
        dist.broadcast(h, src=src, group=group)
        if dist.get_rank() != src:
            t.copy_(h)
    else:
        dist.broadcast(t, src=src, group=group)


def slice_bounds(total, world, rank):
    """Contiguous, balanced slice [lo, hi) of `total` items for `rank`."""
    base, extra = divmod(int(total), int(world))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def plan_bucket_ranges(counts, world):
    """Split the 256 leading-string buckets into `world` contiguous ranges of
    about equal suffix count: range g ends at the first bucket boundary at or
    past g+1 shares of the total.  Returns [(lo, hi)] (hi exclusive); a range may
    be empty when one bucket outweighs a share."""
    counts = np.asarray(counts, np.uint64)
    assert counts.shape == (256,)
    cum = np.concatenate([[0], np.cumsum(counts)]).astype(np.float64)
    total = cum[-1]
    cuts = [0]
    for g in range(1, world):
        target = total * g / world
        b = int(np.searchsorted(cum, target, side="left"))
        # the nearer of the two boundaries around the target
        if b > 0 and abs(cum[b - 1] - target) <= abs(cum[min(b, 256)] - target):
            b -= 1
        cuts.append(min(max(b, cuts[-1]), 256))
    cuts.append(256)
    return [(cuts[g], cuts[g + 1]) for g in range(world)]


def shard_rows(counts, ranges):
    """(row_begin, rows) of every shard: row 0 is the end-of-text suffix."""
    counts = np.asarray(counts, np.uint64)
    cum = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    return [(1 + int(cum[lo]), int(cum[hi] - cum[lo])) for lo, hi in ranges]


class NeedsWholeText(RuntimeError):
    """A shard met ties that only a whole-text build orders."""


def build_sa_sharded(n, counts, build_range, sa, bw, group=None):
    """Fills sa (n+1 rows) and bw (n+1 bytes; bw[r] = BWT symbol of row r, r >= 1)
    on every rank.  counts: the 256 bucket counts (same on every rank).
    build_range(lo, hi, rows_out, bw_out) -> (row_begin, rows, endloc) sorts one
    shard into the front of the two views it is given, or raises
    NeedsWholeText.  Returns endloc.  Collectives: one 4-word all-reduce, then one
    all-gather per array (or one broadcast per shard when shards are lopsided)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    ranges = plan_bucket_ranges(counts, world)
    rows_of = shard_rows(counts, ranges)
    lo, hi = ranges[rank]
    begin, rows = rows_of[rank]
    # needs-whole-text ranks, endloc, rows built, failed ranks: every rank reaches
    # the all-reduce whatever happened to its shard, so that no peer waits forever
    status = torch.zeros(4, dtype=torch.int64, device="cpu" if _host_staged(group) else sa.device)
    failure = None
    if rows:
        try:
            b, r, endloc = build_range(lo, hi, sa[begin:begin + rows], bw[begin:begin + rows])
            if (b, r) != (begin, rows):
                raise RuntimeError(f"shard [{lo},{hi}) came out as rows {b}+{r}, planned {begin}+{rows}")
            status[1] = endloc
            status[2] = r
        except NeedsWholeText:
            status[0] = 1
        except Exception as e:  # noqa: BLE001 - re-raised below, after the collective
            failure = e
            status[3] = 1
    dist.all_reduce(status, op=dist.ReduceOp.SUM, group=group)
    if failure is not None:
        raise failure
    if int(status[3]):
        raise RuntimeError(f"{int(status[3])} rank(s) failed to build their shard")
    if int(status[0]):
        raise NeedsWholeText("a shard holds long repeats; build this text whole on each rank")
    if int(status[2]) != n:
        raise RuntimeError(f"shards hold {int(status[2])} rows, the text has {n} suffixes")
    sa[0] = n
    _exchange_shards(sa, rows_of, rank, group)
    _exchange_shards(bw, rows_of, rank, group)
    return int(status[1])


def _exchange_shards(arr, rows_of, rank, group):
    """Every rank holds its own shard arr[begin:begin+rows]; afterwards all hold
    all.  Shards of about equal size (the planned case) go through ONE all-gather
    of shards padded to the longest -- a ring over xGMI moves it at full link
    rate -- and are copied into place; lopsided ones (a bucket that outweighs a
    share) are broadcast one by one so that no padding travels."""
    world = len(rows_of)
    longest = max(r for _, r in rows_of)
    total = sum(r for _, r in rows_of)
    if world > 1 and longest * world <= total + total // 4 + 1024:
        begin, rows = rows_of[rank]
        send = torch.empty(longest, dtype=arr.dtype, device=arr.device)
        send[:rows] = arr[begin:begin + rows]
        recv = torch.empty(longest * world, dtype=arr.dtype, device=arr.device)
        if arr.is_cuda and _host_staged(group):
            hs, hr = send.cpu(), torch.empty(longest * world, dtype=arr.dtype)
            dist.all_gather_into_tensor(hr, hs, group=group)
            recv.copy_(hr)
        else:
            dist.all_gather_into_tensor(recv, send, group=group)
        for g, (b, r) in enumerate(rows_of):
            if r and g != rank:
                arr[b:b + r] = recv[g * longest:g * longest + r]
        return
    for g, (b, r) in enumerate(rows_of):
        if r:
            src = dist.get_global_rank(group, g) if group is not None else g
            _broadcast(arr[b:b + r], src, group)


def search_sharded(R, search_slice, gather_to=None, group=None):
    """Rank g searches reads [lo_g, hi_g) with search_slice(lo, hi) ->
    (sp, ep, pos) tensors of hi-lo entries.  With gather_to = a rank, that rank
    gets the three arrays for all R reads in read order (others get None);
    otherwise every rank keeps its slice: (lo, hi, sp, ep, pos)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    lo, hi = slice_bounds(R, world, rank)
    sp, ep, pos = search_slice(lo, hi)
    if gather_to is None:
        return lo, hi, sp, ep, pos
    out = []
    for t in (sp, ep, pos):
        full = torch.empty(R, dtype=t.dtype, device=t.device)
        for g in range(world):
            glo, ghi = slice_bounds(R, world, g)
            if ghi > glo:
                if g == rank:
                    full[glo:ghi] = t
                src = dist.get_global_rank(group, g) if group is not None else g
                _broadcast(full[glo:ghi], src, group)
        out.append(full if rank == gather_to else None)
    return tuple(out)
