"""Window sharding across the GPUs of one node.

Windows are independent (there is no exchange step inside the optimisation), so
the only collective is one all-gather of the refined global poses; inputs (heat-maps above all) are
never scattered from a root: every rank loads or synthesises the frames of its own contiguous window
range.  One process per GPU, `torch.distributed` with backend "nccl" (= RCCL over xGMI) on the GPU box,
"gloo" in the CPU tests.
"""
import numpy as np
import torch
import torch.distributed as dist


def shard_range(n_windows, rank, world):
    """Contiguous [lo, hi) range of rank `rank`: sizes differ by at most one, lower ranks get the extra."""
    base, extra = divmod(n_windows, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_indices(n_windows, rank, world, block=None):
    """Window indices of rank `rank`.  block=None: the contiguous range of `shard_range`.  block=k: block-cyclic -- blocks
    of k consecutive windows are dealt round-robin over the ranks, so that stretches of a long sequence whose windows finish
    early (few L-BFGS evaluations) or late are spread over all ranks instead of landing on one; a
    block keeps k overlapping windows = 8k + 2 contiguous frames together, so frames are still stored (almost) once."""
    if block is None:
        lo, hi = shard_range(n_windows, rank, world)
        return np.arange(lo, hi)
    blocks = np.arange(0, n_windows, block)
    mine = blocks[rank::world]
    return np.concatenate([np.arange(b, min(b + block, n_windows)) for b in mine]) if len(mine) else np.zeros(0, dtype=np.int64)


def frame_span(starts, lo, hi, seq_len):
    """Frames [f0, f1) that the windows [lo, hi) touch: what a rank has to hold in HBM."""
    if hi <= lo:
        return 0, 0
    return int(starts[lo]), int(starts[hi - 1]) + seq_len


def frame_runs(starts, idx, seq_len):
    """Frames the windows `idx` (ascending window indices into `starts`) touch, as the shortest list of contiguous runs
    [(f0, f1), ...], and every window's first frame re-indexed into the concatenation of those runs -- what a rank holds in HBM
    for a block-cyclic shard of ONE long sequence: each block of k overlapping windows is 8k + 2 contiguous frames, stored once;
    only the `seq_len - stride` halo frames between blocks that went to different ranks exist twice, once on each."""
    runs, local0 = [], np.zeros(len(idx), dtype=np.int32)
    base = 0
    for n, i in enumerate(idx):
        f0, f1 = int(starts[i]), int(starts[i]) + seq_len
        if runs and f0 <= runs[-1][1]:                       # overlaps / touches the current run: extend it
            runs[-1][1] = max(runs[-1][1], f1)
        else:
            if runs:
                base += runs[-1][1] - runs[-1][0]
            runs.append([f0, f1])
        local0[n] = base + f0 - runs[-1][0]
    return [tuple(r) for r in runs], local0


def all_gather_windows(local, n_windows, group=None):
    """local [n_local, ...] (any float dtype, same trailing shape on every rank) -> [n_windows, ...] on
    every rank, in window order.  Shards may be ragged by one window; they are padded for the collective."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    sizes = [shard_range(n_windows, r, world) for r in range(world)]
    cap = max(hi - lo for lo, hi in sizes)
    lo, hi = sizes[rank]
    if local.shape[0] != hi - lo:
        raise ValueError("rank %d holds %d windows, its shard is %d" % (rank, local.shape[0], hi - lo))
    # gloo (CPU tests, single-GPU rehearsals) gathers host copies; nccl = RCCL gathers device buffers over xGMI
    dev = local.device
    shape = (cap,) + tuple(local.shape[1:])
    if dist.get_backend(group) == "gloo":
        buf = torch.zeros(shape, dtype=local.dtype, device="cpu")
        buf[: hi - lo] = local
        out = [torch.empty_like(buf) for _ in range(world)]
        dist.all_gather(out, buf, group=group)
        return torch.cat([o[: h - l] for o, (l, h) in zip(out, sizes)], dim=0).to(dev)
    # one flat receive buffer, one collective; equal shards (the weak-scaling bench) need no copy afterwards
    if hi - lo == cap:
        buf = local.contiguous()
    else:
        buf = torch.zeros(shape, dtype=local.dtype, device=dev)
        buf[: hi - lo] = local
    out = torch.empty((world * cap,) + tuple(local.shape[1:]), dtype=local.dtype, device=dev)
    dist.all_gather_into_tensor(out, buf, group=group)
    if all(h - l == cap for l, h in sizes):
        return out
    return torch.cat([out[r * cap: r * cap + (h - l)] for r, (l, h) in enumerate(sizes)], dim=0)


def all_gather_indexed(local, n_windows, block, group=None):
    """all_gather for `shard_indices(..., block)` shards: every rank contributes its rows, every rank receives all
    n_windows rows in window order.  One collective of equal-size (padded) shards, then one scatter by index."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    idx = [shard_indices(n_windows, r, world, block) for r in range(world)]
    if local.shape[0] != len(idx[rank]):
        raise ValueError("rank %d holds %d windows, its shard is %d" % (rank, local.shape[0], len(idx[rank])))
    cap = max(len(i) for i in idx)
    dev = local.device
    gloo = dist.get_backend(group) == "gloo"
    buf = torch.zeros((cap,) + tuple(local.shape[1:]), dtype=local.dtype, device="cpu" if gloo else dev)
    buf[: local.shape[0]] = local
    if gloo:
        parts = [torch.empty_like(buf) for _ in range(world)]
        dist.all_gather(parts, buf, group=group)
    else:
        flat = torch.empty((world * cap,) + tuple(local.shape[1:]), dtype=local.dtype, device=dev)
        dist.all_gather_into_tensor(flat, buf, group=group)
        parts = [flat[r * cap:(r + 1) * cap] for r in range(world)]
    out = torch.empty((n_windows,) + tuple(local.shape[1:]), dtype=local.dtype, device=parts[0].device)
    for r in range(world):
        if len(idx[r]):
            out[torch.as_tensor(idx[r], device=out.device)] = parts[r][: len(idx[r])]
    return out.to(dev)


def optimize_sharded(run_shard, starts, seq_len=10, group=None):
    """Run `run_shard(lo, hi, f0, f1)` (-> tensor [hi-lo, T, 15, 3]) on this rank's window range and
    all-gather the refined poses.  `starts` [n_windows] first frame of each window (same on all ranks)."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    n = len(starts)
    lo, hi = shard_range(n, rank, world)
    f0, f1 = frame_span(np.asarray(starts), lo, hi, seq_len)
    local = run_shard(lo, hi, f0, f1)
    return all_gather_windows(local, n, group)
