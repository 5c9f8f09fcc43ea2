"""Several GPUs, one process each (torch.distributed; backend "nccl" is RCCL).

Search: the index is replicated, reads are split into contiguous slices, no
exchange on the data path; results are gathered in read order only when the
caller asks for them on one rank.

Index build: the text is replicated (a quarter byte per base), every rank sorts
the suffixes of its own range of leading four-base strings -- a contiguous
range of suffix-array rows -- and derives from its rows what the index needs
of them: the BWT symbols packed to 2 bits and every 8th row.  Those index
products (0.75 bytes per base) are what the ranks all-gather
(build_index_parts_sharded); each rank then lays out bit-planes and rank
blocks for the whole text.  The rows themselves (4 bytes per base) travel only
when a rank wants the whole suffix array for its search-only position records
(build_sa_sharded), after the index is usable.  Ids are never exchanged before
the sort: each rank reads its suffixes straight off its copy of the text.

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
    sa[0] = n - (1 << 32) if n >= (1 << 31) and sa.dtype == torch.int32 else n  # uint32 bits in an int32 tensor
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


def _all_gather_var(send, sizes, group):
    """Every rank contributes send[:sizes[rank]] (1-D, same dtype on all ranks).
    Returns (buffer, offsets): rank g's part is buffer[offsets[g] : offsets[g] +
    sizes[g]].  Parts of about equal size go through ONE all-gather of slots
    padded to the longest; lopsided ones are broadcast one by one so that no
    padding travels."""
    world = len(sizes)
    rank = dist.get_rank(group)
    longest = max(sizes)
    total = sum(sizes)
    dev = send.device
    staged = send.is_cuda and _host_staged(group)
    if world > 1 and longest * world <= total + total // 4 + 1024:
        slot = torch.zeros(longest, dtype=send.dtype, device=dev)
        slot[:sizes[rank]] = send[:sizes[rank]]
        recv = torch.empty(longest * world, dtype=send.dtype, device=dev)
        if staged:
            hr = torch.empty(longest * world, dtype=send.dtype)
            dist.all_gather_into_tensor(hr, slot.cpu(), group=group)
            recv.copy_(hr)
        else:
            dist.all_gather_into_tensor(recv, slot, group=group)
        return recv, [g * longest for g in range(world)]
    recv = torch.empty(max(total, 1), dtype=send.dtype, device=dev)
    offsets, at = [], 0
    for g in range(world):
        offsets.append(at)
        if sizes[g]:
            part = recv[at:at + sizes[g]]
            if g == rank:
                part.copy_(send[:sizes[g]])
            src = dist.get_global_rank(group, g) if group is not None else g
            _broadcast(part, src, group)
        at += sizes[g]
    return recv, offsets


class ShardedParts:
    """What build_index_parts_sharded leaves on every rank."""

    def __init__(self, endloc, rows_of, bw2, bw2_offsets, samples):
        self.endloc = endloc            # row of suffix 0
        self.rows_of = rows_of          # [(row_begin, rows)] per shard
        self.bw2 = bw2                  # uint8: every shard's BWT symbols, 2 bits each
        self.bw2_offsets = bw2_offsets  # byte offset of shard g in bw2
        self.samples = samples          # int32[n // 8 + 1]: SA[8 i]


def sample_span(begin, rows):
    """Rows r of [begin, begin + rows) with r % 8 == 0: (offset of the first one
    inside the shard, index of its sample, how many)."""
    first = (-begin) % 8
    if first >= rows:
        return first, (begin + first) // 8, 0
    return first, (begin + first) // 8, (rows - first + 7) // 8


def build_index_parts_sharded(n, counts, build_range, pack_bw2, rows_buf, bw_buf, group=None):
    """The exchange of a sharded index build.  counts: the 256 bucket counts (same
    on every rank).  build_range(lo, hi, rows_out, bw_out) -> (row_begin, rows,
    endloc) sorts this rank's shard into the front of rows_buf / bw_buf (which
    need room for this rank's shard only), or raises NeedsWholeText.
    pack_bw2(bw, rows) -> uint8 tensor of (rows + 3) // 4 bytes, symbol j at bits
    2 (j & 3) of byte j >> 2.  Collectives: one 4-word all-reduce (status), one
    all-gather of packed symbols (n/4 bytes in all), one of samples (n/2 bytes).
    Returns ShardedParts; rows_buf[:rows] still holds this rank's rows."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    ranges = plan_bucket_ranges(counts, world)
    rows_of = shard_rows(counts, ranges)
    lo, hi = ranges[rank]
    begin, rows = rows_of[rank]
    status = torch.zeros(4, dtype=torch.int64, device="cpu" if _host_staged(group) else rows_buf.device)
    failure = None
    if rows:
        try:
            b, r, endloc = build_range(lo, hi, rows_buf[:rows], bw_buf[:rows])
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
    # this rank's index products: 2-bit symbols of its rows, every 8th row
    bw2_sizes = [(r + 3) // 4 for _, r in rows_of]
    mine = pack_bw2(bw_buf[:rows], rows) if rows else torch.empty(0, dtype=torch.uint8, device=bw_buf.device)
    bw2, bw2_offsets = _all_gather_var(mine, bw2_sizes, group)
    spans = [sample_span(b, r) for b, r in rows_of]
    first, _, cnt = spans[rank]
    mine_s = rows_buf[first:rows:8].contiguous() if cnt else torch.empty(0, dtype=rows_buf.dtype,
                                                                           device=rows_buf.device)
    got, offs = _all_gather_var(mine_s, [c for _, _, c in spans], group)
    samples = torch.empty(n // 8 + 1, dtype=rows_buf.dtype, device=rows_buf.device)
    samples[0] = n - (1 << 32) if n >= (1 << 31) and rows_buf.dtype == torch.int32 else n  # SA[0] = n, as uint32 bits
    for g, (_, i0, c) in enumerate(spans):
        if c:
            samples[i0:i0 + c] = got[offs[g]:offs[g] + c]
    return ShardedParts(int(status[1]), rows_of, bw2, bw2_offsets, samples)


def gather_suffix_array(n, rows_of, rows_buf, sa, group=None):
    """The second, optional exchange: every rank's rows (rows_buf[:rows]) into
    sa[0 .. n] on every rank."""
    rank = dist.get_rank(group)
    begin, rows = rows_of[rank]
    sa[0] = n - (1 << 32) if n >= (1 << 31) and sa.dtype == torch.int32 else n
    sa[begin:begin + rows] = rows_buf[:rows]
    _exchange_shards(sa, rows_of, rank, group)


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
