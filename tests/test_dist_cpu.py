"""world_size-2 gloo tests of the sharding + all-gather path (CPU)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from globalegomocap_amd.dist import shard_range, shard_indices, frame_span, all_gather_windows, all_gather_indexed, optimize_sharded
from globalegomocap_amd.sequence import window_starts, merge_batches


def test_shard_ranges_cover_everything_once():
    for n in (0, 1, 7, 12, 240, 241, 65536):
        for world in (1, 2, 3, 8):
            r = [shard_range(n, k, world) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(r, r[1:]))
            sizes = [hi - lo for lo, hi in r]
            assert max(sizes) - min(sizes) <= 1


def test_frame_span_is_contiguous_and_minimal():
    starts = np.concatenate([c * 100 + window_starts(100) for c in range(3)])
    lo, hi = shard_range(len(starts), 1, 2)
    f0, f1 = frame_span(starts, lo, hi, 10)
    assert f0 == starts[lo] and f1 == starts[hi - 1] + 10
    assert frame_span(starts, 5, 5, 10) == (0, 0)


def test_block_cyclic_shards_cover_everything_once():
    for n in (0, 1, 15, 16, 17, 1563, 12499):
        for world in (1, 2, 8):
            for block in (None, 1, 8, 16):
                idx = [shard_indices(n, r, world, block) for r in range(world)]
                allidx = np.sort(np.concatenate(idx)) if n else np.zeros(0)
                assert np.array_equal(allidx, np.arange(n))
                if block is not None and n >= world * block:
                    sizes = [len(i) for i in idx]
                    assert max(sizes) - min(sizes) <= block


def test_frame_runs_of_block_cyclic_shards():
    """What a rank holds of ONE long sequence: the frames of its blocks, stored once, windows re-indexed into that buffer."""
    from globalegomocap_amd.dist import frame_runs
    n, T = 12499, 10
    starts = 8 * np.arange(n)
    for world, block in ((1, 64), (2, 64), (8, 64), (8, 8), (8, 1), (3, 5)):
        held = 0
        for r in range(world):
            idx = shard_indices(n, r, world, block)
            runs, local0 = frame_runs(starts, idx, T)
            keep = np.concatenate([np.arange(a, b) for a, b in runs]) if runs else np.zeros(0, dtype=np.int64)
            assert all(b > a for a, b in runs) and all(runs[k][0] > runs[k - 1][1] for k in range(1, len(runs)))     # disjoint, ascending, maximal
            for m, i in enumerate(idx[:: max(1, len(idx) // 50)]):
                k = list(idx).index(i) if len(idx) < 2000 else int(np.searchsorted(idx, i))
                assert np.array_equal(keep[local0[k]:local0[k] + T], np.arange(starts[i], starts[i] + T))
            held += len(keep)
        n_frames = 8 * (n - 1) + T
        n_blocks = (n + block - 1) // block
        # every frame once, plus the 2-frame halo at each boundary between blocks of different ranks
        assert n_frames <= held <= n_frames + 2 * (n_blocks - 1) * (1 if world > 1 else 0)
    assert frame_runs(starts, np.zeros(0, dtype=np.int64), T)[0] == []


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_windows, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        starts = np.concatenate([c * 100 + window_starts(100) for c in range(2)])[:n_windows]
        full = torch.arange(n_windows * 10 * 15 * 3, dtype=torch.float64).reshape(n_windows, 10, 15, 3)

        def run_shard(lo, hi, f0, f1):
            # stand-in for WindowEngine.optimize_windows on this rank's frames: returns its rows of `full`
            assert (f0, f1) == frame_span(starts, lo, hi, 10)
            return full[lo:hi].clone()

        out = optimize_sharded(run_shard, starts)
        ok = torch.equal(out, full)
        # ragged float32 shard through the raw gather as well
        lo, hi = shard_range(n_windows, rank, world)
        g32 = all_gather_windows(full[lo:hi].float(), n_windows)
        ok = ok and torch.equal(g32, full.float())
        # block-cyclic shards (load balancing over a long sequence): same rows, window order restored by the gather
        mine = shard_indices(n_windows, rank, world, block=4)
        gi = all_gather_indexed(full[torch.as_tensor(mine, dtype=torch.long)].clone(), n_windows, block=4)
        ok = ok and torch.equal(gi, full)
        merged = merge_batches(out[:12].numpy())
        q.put((rank, bool(ok), merged.shape))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_windows", [24, 23, 1])
def test_two_rank_gather_reassembles_window_order(n_windows):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_windows, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, shape in res:
        assert ok, rank
        if n_windows >= 12:
            assert shape == (98, 15, 3)
