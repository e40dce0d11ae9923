"""Sharding a batch of independent blocks over the GPUs of one node.

The reference has no distributed code; the only parallelism is new:
messages are independent, and blocks of one message become independent once
the plain-SHA chain pre-pass has produced their pre-states (lib.rs:188,236).
So ranks take contiguous block ranges, there is NO data-path collective, and
the one positional quantity -- SpreadConfig.num_limb_sum (spread.rs:26) -- is
closed-form: block j starts at cursor0 + j * limb_calls_per_block.

The optional all-gather (north_star: "RCCL all-gather of per-block witness
columns over xGMI") only assembles the already-computed shards on every rank;
it works on any torch.distributed backend (nccl = RCCL on the GPU box, gloo in
the CPU tests).
"""


def shard_range(n_items, world, rank):
    """Contiguous [start, start+count) of rank; the remainder goes to the first ranks."""
    base, rem = divmod(n_items, world)
    start = rank * base + min(rank, rem)
    return start, base + (1 if rank < rem else 0)


def shard_cursor(cursor0, start_block, limb_calls_per_block):
    """num_limb_sum at which a shard starting at block `start_block` begins."""
    return cursor0 + start_block * limb_calls_per_block


def shard_row_window(cursor0, start_block, n_blocks, limb_calls_per_block, ncols):
    """(first absolute chip row, number of rows) a shard's column buffers cover:
    buffer row 0 is absolute row shard_cursor // ncols (hsw.h hsw_chip_rows)."""
    c = shard_cursor(cursor0, start_block, limb_calls_per_block)
    rows = (c % ncols + limb_calls_per_block * n_blocks + ncols - 1) // ncols
    return c // ncols, rows


def allgather_gate(dist, local_gate, counts, cells_per_block, group=None):
    """All-gather the gate stream.  counts[r] = blocks of rank r.  Shards may be
    uneven: every rank pads to the largest shard, the pad is dropped after the
    gather.  Returns the concatenated stream (sum(counts) * cells_per_block, 4)."""
    import torch
    world = len(counts)
    cmax = max(counts)
    pad_rows = cmax * cells_per_block
    if local_gate.shape[0] != pad_rows:
        buf = torch.zeros((pad_rows,) + tuple(local_gate.shape[1:]), dtype=local_gate.dtype,
                          device=local_gate.device)
        buf[: local_gate.shape[0]] = local_gate
    else:
        buf = local_gate.contiguous()
    out = torch.empty((world * pad_rows,) + tuple(local_gate.shape[1:]), dtype=local_gate.dtype,
                      device=local_gate.device)
    dist.all_gather_into_tensor(out, buf, group=group)
    if all(c == cmax for c in counts):
        return out
    parts = [out[r * pad_rows: r * pad_rows + counts[r] * cells_per_block] for r in range(world)]
    return torch.cat(parts, dim=0)


def allgather_chip(dist, local_cols, cursor0, starts, counts, limb_calls_per_block, ncols, group=None):
    """All-gather one chip column family (dense or spread).

    local_cols: (ncols, rows_r, 4) -- this rank's column buffers, buffer row 0 =
    absolute row shard_cursor_r // ncols.  Returns (ncols, total_rows, 4) with
    buffer row 0 = absolute row cursor0 // ncols.  When a shard boundary falls
    inside a row (limb count not a multiple of ncols) the two neighbouring
    ranks each own some cells of that row; ownership follows limb call
    n = row * ncols + column, exactly like spread.rs:202-231."""
    import torch
    world = len(counts)
    windows = [shard_row_window(cursor0, starts[r], counts[r], limb_calls_per_block, ncols)
               for r in range(world)]
    rmax = max(w[1] for w in windows)
    buf = torch.zeros((ncols, rmax, local_cols.shape[2]), dtype=local_cols.dtype, device=local_cols.device)
    buf[:, : local_cols.shape[1]] = local_cols
    flat = torch.empty((world * ncols,) + tuple(buf.shape[1:]), dtype=buf.dtype, device=buf.device)
    dist.all_gather_into_tensor(flat, buf.contiguous(), group=group)     # concatenates along dim 0
    gathered = flat.view((world, ncols) + tuple(buf.shape[1:]))
    row0 = cursor0 // ncols
    total_limbs = limb_calls_per_block * sum(counts)
    total_rows = (cursor0 % ncols + total_limbs + ncols - 1) // ncols
    out = torch.zeros((ncols, total_rows, local_cols.shape[2]), dtype=local_cols.dtype,
                      device=local_cols.device)
    for r in range(world):
        if counts[r] == 0:
            continue
        first = shard_cursor(cursor0, starts[r], limb_calls_per_block)       # first limb of rank r
        last = first + counts[r] * limb_calls_per_block - 1
        wrow0, _ = windows[r]
        for c in range(ncols):
            if last < c:
                continue
            lo = (first + ncols - 1 - c) // ncols        # first row with row*ncols + c >= first
            hi = (last - c) // ncols
            if hi < lo:
                continue
            out[c, lo - row0: hi - row0 + 1] = gathered[r, c, lo - wrow0: hi - wrow0 + 1]
    return out


def allgather_seeds(dist, local_blocks, local_pre_states, counts, group=None):
    """All-gather the 96-byte *seeds* of every block (64 message bytes + 8 pre-state
    words) instead of its 2.39 MB of witness cells.  Every rank can then expand
    any block range locally (a witness stream is a pure function of its seed and
    of the closed-form chip cursor), which on xGMI is ~25,000x less traffic than
    gathering the columns and, with 288 GB of HBM per GPU, lets every GPU hold
    the complete 8-GPU witness (65,536 blocks = 156 GB) if it needs to.
    local_blocks: (n_r, 64) uint8; local_pre_states: (n_r, 8) int32; counts[r] = n_r.
    Returns (blocks (sum n, 64) uint8, pre_states (sum n, 8) int32) in rank order."""
    import torch
    world = len(counts)
    cmax = max(counts)
    dev = local_blocks.device
    buf = torch.zeros((cmax, 96), dtype=torch.uint8, device=dev)
    n = local_blocks.shape[0]
    if n:
        buf[:n, :64] = local_blocks
        buf[:n, 64:] = local_pre_states.contiguous().view(torch.uint8).reshape(n, 32)
    out = torch.empty((world * cmax, 96), dtype=torch.uint8, device=dev)
    dist.all_gather_into_tensor(out, buf, group=group)
    parts = [out[r * cmax: r * cmax + counts[r]] for r in range(world)]
    allseeds = torch.cat(parts, dim=0) if not all(c == cmax for c in counts) else out
    blocks = allseeds[:, :64].contiguous()
    pre = allseeds[:, 64:].contiguous().view(torch.int32).reshape(-1, 8)
    return blocks, pre
