"""Whole-sequence driver: every chunk directory of a sequence in ONE batched device call.

The reference's `optimize_whole_sequence.py` walks the chunk directories of a sequence (natsorted), calls
`main()` on each -- 12 windows, one after the other -- and averages the per-chunk error dicts
(`optimize_whole_sequence.py:48-118`).  Here the chunks are read by background threads (which also start the
heat-map uploads), all their windows go through `SequenceOptimizer.run` together (BASELINE configs[1]: a
2000-frame sequence = 20 chunks = 240 windows per call), and the per-chunk merge / smoothing / error report run
on the device as well.  Results, keys, printed summary and the order in which the reparameterisation noise is
drawn (chunk by chunk, window by window, local then global) follow the reference.

    python -m globalegomocap_amd.whole_sequence --data_path data/jian3
"""
import os
import pickle
import re
import threading
from collections import OrderedDict

import numpy as np
import torch

from .optimizer import SequenceOptimizer, GLOBAL_VAE_PATH, LOCAL_VAE_PATH
from .sequence import SEQ_LEN, OVERLAP, window_starts, cut_windows, merge_batches, relative_global_numpy, to_global_numpy

SUMMARY_LINES = (          # (label printed by the reference, key) in print order, None = separator
    ("Average original global pose mpjpe", "original_global_mpjpe"), ("Average mid global pose mpjpe", "mid_global_mpjpe"),
    ("Average optimized global pose mpjpe", "optimized_global_mpjpe"), None,
    ("Average original cam pose error", "original_camera_pos_error"), ("Average optimized cam pose error", "optimized_camera_pos_error"), None,
    ("Average original aligned cam pose error", "original_aligned_camera_pos_error"),
    ("Average optimized aligned cam pose error", "optimized_aligned_camera_pos_error"), None,
    ("Average original_aligned_global_mpjpe", "original_aligned_global_mpjpe"), ("Average aligned_mid_seq_mpjpe", "aligned_mid_seq_mpjpe"),
    ("Average optimized_aligned_global_mpjpe", "optimized_aligned_global_mpjpe"), None,
    ("Average aligned original global pose mpjpe", "aligned_original_mpjpe"),
    ("Average aligned mid local pose mpjpe", "aligned_mid_optimized_mpjpe"),
    ("Average aligned optimized global pose mpjpe", "aligned_optimized_mpjpe"), None,
    ("Average bone length aligned original global pose mpjpe", "bone_length_aligned_original_mpjpe"),
    ("Average bone length aligned mid local pose mpjpe", "bone_length_aligned_mid_optimized_mpjpe"),
    ("Average bone length aligned optimized global pose mpjpe", "bone_length_aligned_optimized_mpjpe"), None,
)


_reader_local = threading.local()      # per reader thread: copy stream + pinned staging buffer, reused from chunk to chunk


def natural_key(name):
    """Sort key equivalent to natsort.natsorted (default algorithm: case-sensitive text, unsigned integers) for directory
    names like chunk_2 < chunk_10 (optimize_whole_sequence.py:48)."""
    return [int(t) if t.isdigit() else t for t in re.split(r"(\d+)", name)]


def list_chunks(data_dir):
    """Chunk directories in the reference's order (`natsorted(os.listdir(data_dir))`, directories only)."""
    names = sorted(os.listdir(data_dir), key=natural_key)
    return [os.path.join(data_dir, n) for n in names if os.path.isdir(os.path.join(data_dir, n))]


def load_chunk(path, device=None):
    """`<chunk>/test_data.pkl` (optimizer.py:315-324) as dense arrays; KeyError on a missing key like the reference.
    With `device`, the heat-maps (99 % of the bytes) go straight to that device from the calling thread."""
    with open(os.path.join(path, "test_data.pkl"), "rb") as f:
        d = pickle.load(f)
    c = {"path": path,
         "est_local": np.asarray(d["estimated_local_skeleton"], dtype=np.float64),
         "gt": np.asarray(d["gt_global_skeleton"], dtype=np.float64),
         "cams": np.asarray(d["camera_pose_list"], dtype=np.float64)}
    heat = d["heatmap_list"]
    if device is None:
        c["heat"] = np.asarray(heat, dtype=np.float32)
    else:
        # into this reader thread's pinned staging buffer (one np.stack-free pass, no intermediate pageable copy), then one
        # async H2D copy on the thread's own stream; the consumer waits on the event and marks the tensor as used on its stream
        n = len(heat)
        shape = (n,) + tuple(np.shape(heat[0])) if n else (0, 64, 64, 15)
        tl = _reader_local
        if getattr(tl, "stream", None) is None or tl.device != device:
            tl.stream, tl.device, tl.stage, tl.copied = torch.cuda.Stream(device=device), device, None, None
        numel = int(np.prod(shape))
        if tl.stage is None or tl.stage.numel() < numel:
            tl.stage = torch.empty(max(numel, 1), dtype=torch.float32).pin_memory()
        if tl.copied is not None:
            tl.copied.synchronize()                        # the previous chunk's copy has left the staging buffer
        view = tl.stage[:numel].view(shape).numpy()
        if n:
            np.stack(heat, out=view) if isinstance(heat, (list, tuple)) else np.copyto(view, np.asarray(heat, dtype=np.float32))
        with torch.cuda.stream(tl.stream):
            c["heat"] = tl.stage[:numel].view(shape).to(device, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(tl.stream)
        tl.copied = ev
        c["heat_ready"] = ev
    return c


class ChunkStream:
    """Reads chunk pickles on `workers` background threads, at most `depth` chunks ahead of the consumer, and yields
    them in directory order."""

    def __init__(self, paths, depth=8, workers=4, device=None):
        from concurrent.futures import ThreadPoolExecutor
        self._paths = list(paths)
        self._depth = max(1, depth)
        self._device = device
        self._pool = ThreadPoolExecutor(max_workers=max(1, workers))

    def __iter__(self):
        pending = []
        it = iter(self._paths)
        try:
            while True:
                while len(pending) < self._depth:
                    p = next(it, None)
                    if p is None:
                        break
                    pending.append(self._pool.submit(load_chunk, p, self._device))
                if not pending:
                    return
                yield pending.pop(0).result()          # a reader's exception (e.g. KeyError) surfaces here
        finally:
            for f in pending:
                f.cancel()
            self._pool.shutdown(wait=True)


def _batches(stream, chunks_per_batch):
    batch = []
    for c in stream:
        batch.append(c)
        if chunks_per_batch and len(batch) == chunks_per_batch:
            yield batch
            batch = []
    if batch:
        yield batch


def optimize_sequences(data_dirs, camera_model_path, vae_weight=0.0, gmm_weight=0.0, smoothness_weight=0.001,
                       bone_length_weight=0.01, weight_3d=0.01, reproj_weight=0.01, final_smooth=True, merge=True,
                       global_vae_path=GLOBAL_VAE_PATH, local_vae_path=LOCAL_VAE_PATH, chunks_per_batch=None, optimizer=None,
                       device_metrics=True, verbose=True, seq_len=SEQ_LEN, overlap=OVERLAP):
    """Several sequences in ONE batched device call (BASELINE configs[2]: all test sequences concurrently on one GPU):
    the chunks of every directory of `data_dirs` go through the optimiser together, the reports are per sequence.
    Returns a list of (summary, per-chunk error dicts, estimated_pose, optimized_pose, gt_pose), one per directory, each
    exactly what `optimize_directory` returns; the noise is drawn sequence by sequence, chunk by chunk."""
    del gmm_weight, merge                       # accepted and unused, as in the reference
    paths, group_of = [], {}
    for gi, d in enumerate(data_dirs):
        ps = list_chunks(d)
        if not ps:
            raise FileNotFoundError("no chunk directories under %s" % d)
        for q in ps:
            group_of[q] = gi
        paths += ps
    n_groups = len(data_dirs)
    opt = optimizer
    results, est_all, opt_all, gt_all = ([[] for _ in range(n_groups)] for _ in range(4))
    device = torch.device("cuda", torch.cuda.current_device())
    for batch in _batches(ChunkStream(paths, device=device), chunks_per_batch):
        starts, chunk_of, bounds, f_off, eps = [], [], [], 0, []
        for ci, c in enumerate(batch):
            if verbose:
                print("running data: {}".format(c["path"]))
            s = window_starts(len(c["est_local"]), seq_len, overlap)
            c["starts"] = s
            starts.append(s + f_off)
            chunk_of.append(np.full(len(s), ci, dtype=np.int64))
            bounds.append((f_off, f_off + len(c["est_local"])))
            f_off += len(c["est_local"])
        n_win = int(sum(len(s) for s in starts))
        if opt is None:
            opt = SequenceOptimizer(camera_model_path, global_vae_path, local_vae_path, max_windows=max(n_win, 1), seq_len=seq_len)
        if n_win > opt.engine.max_windows:
            raise ValueError("%d windows in one batch exceed the engine's max_windows=%d: pass chunks_per_batch" %
                             (n_win, opt.engine.max_windows))
        for c in batch:                          # global torch RNG, in the reference's order (D5)
            eps.append(torch.randn(2 * len(c["starts"]), opt.engine.D))
        w_local, w_global = opt.stage_weights(vae_weight, smoothness_weight, bone_length_weight, weight_3d, reproj_weight)
        for c in batch:
            torch.cuda.current_stream().wait_event(c["heat_ready"])
            c["heat"].record_stream(torch.cuda.current_stream())      # allocated on a reader's stream, consumed on this one
        heat_d = batch[0]["heat"] if len(batch) == 1 else torch.cat([c["heat"] for c in batch])
        mid_local, opt_global, _ = opt.run(np.concatenate([c["est_local"] for c in batch]), np.concatenate([c["cams"] for c in batch]),
                                           heat_d, np.concatenate(starts),
                                           np.concatenate(chunk_of), bounds, w_local, w_global, eps=torch.cat(eps), keep_device=True)
        mid_np = mid_local.cpu().numpy()
        w0 = 0
        for c in batch:
            nw = len(c["starts"])
            sl = slice(w0, w0 + nw)
            w0 += nw
            if nw == 0:
                continue
            loc_w, cam_w = cut_windows(c["est_local"], c["starts"], seq_len), cut_windows(c["cams"], c["starts"], seq_len)
            est_seq = merge_batches(to_global_numpy(relative_global_numpy(loc_w, cam_w), cam_w), overlap)
            mid_seq = merge_batches(to_global_numpy(relative_global_numpy(mid_np[sl], cam_w), cam_w), overlap)
            gt_seq = merge_batches(cut_windows(c["gt"], c["starts"], seq_len), overlap)
            e = opt.engine
            if device_metrics:
                opt_seq_d = e.merge_windows(opt_global[sl], 1, overlap=overlap, smooth=bool(final_smooth))
                res = e.calculate_errors(est_seq, mid_seq, opt_seq_d, gt_seq)
                opt_seq = opt_seq_d.cpu().numpy()
            else:
                from .errors import calculate_errors
                from .sequence import final_smooth as _smooth
                opt_seq = merge_batches(opt_global[sl].cpu().numpy(), overlap)
                if final_smooth:
                    opt_seq = _smooth(opt_seq)
                res = calculate_errors(est_seq, mid_seq, opt_seq, gt_seq)
            gi = group_of[c["path"]]
            results[gi].append(res)
            est_all[gi].extend(list(est_seq)); opt_all[gi].extend(list(opt_seq)); gt_all[gi].extend(list(gt_seq))
            if verbose and res["bone_length_aligned_optimized_mpjpe"] > res["bone_length_aligned_mid_optimized_mpjpe"]:
                print(res)
    out = []
    for gi in range(n_groups):
        summary = OrderedDict()
        for k in results[gi][0]:
            summary[k] = (np.mean([r[k] for r in results[gi]], axis=0) if k == "joints_error"
                          else float(np.average([r[k] for r in results[gi]])))
        if verbose:
            if n_groups > 1:
                print("sequence: {}".format(data_dirs[gi]))
            for line in SUMMARY_LINES:
                print("-----------------------------------------" if line is None else "{}: {}".format(line[0], summary[line[1]]))
            print("joints error is: {}".format(summary["joints_error"]))
            print("-------------------------------------------------------------")
        out.append((summary, results[gi], est_all[gi], opt_all[gi], gt_all[gi]))
    return out


def optimize_directory(data_dir, camera_model_path, *args, **kwargs):
    """One sequence = the reference's `optimize_whole_sequence.py`.  Returns (summary OrderedDict, per-chunk error
    dicts, estimated_pose, optimized_pose, gt_pose) -- the three pose lists are the concatenations
    `optimize_whole_sequence.py:65-67` builds.  Arguments as `optimize_sequences`."""
    return optimize_sequences([data_dir], camera_model_path, *args, **kwargs)[0]


def _cli():
    import argparse
    from .camera import DEFAULT_CALIBRATION
    truthy = lambda x: str(x).lower() == "true"          # noqa: E731  (the reference's own flag parser)
    p = argparse.ArgumentParser(description="Data directory number")
    p.add_argument("--data_path", required=True, type=str)
    p.add_argument("--camera", type=str, default=DEFAULT_CALIBRATION)
    p.add_argument("--vae", type=float, default=0.00)
    p.add_argument("--gmm", type=float, default=0.00)
    p.add_argument("--smooth", type=float, default=0.001)
    p.add_argument("--bone_length", type=float, default=0.01)
    p.add_argument("--weight_3d", type=float, default=0.01)
    p.add_argument("--reproj_weight", type=float, default=0.01)
    p.add_argument("--save", default=False, type=truthy)
    p.add_argument("--final_smooth", default=True, type=truthy)
    p.add_argument("--merge", default=True, type=truthy)
    p.add_argument("--chunks_per_batch", type=int, default=None, help="chunks optimised per device call (default: all)")
    a = p.parse_args()
    if a.save:
        raise NotImplementedError("--save writes open3d meshes (optimizer.py:452-504): outside the hot path")
    optimize_directory(a.data_path, a.camera, a.vae, a.gmm, a.smooth, a.bone_length, a.weight_3d, a.reproj_weight,
                       final_smooth=a.final_smooth, merge=a.merge, chunks_per_batch=a.chunks_per_batch)


if __name__ == "__main__":
    _cli()
