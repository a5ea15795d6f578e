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
from .sequence import (SEQ_LEN, OVERLAP, window_starts, cut_windows, merge_batches, merge_chunks, relative_global_numpy,
                       to_global_numpy)

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
_heat_pool = {}                        # device -> frame buffers the readers fill in place (kept between calls)


def natural_key(name):
    """Sort key equivalent to natsort.natsorted (default algorithm: case-sensitive text, unsigned integers) for directory
    names like chunk_2 < chunk_10 (optimize_whole_sequence.py:48)."""
    return [int(t) if t.isdigit() else t for t in re.split(r"(\d+)", name)]


def list_chunks(data_dir):
    """Chunk directories in the reference's order (`natsorted(os.listdir(data_dir))`, directories only)."""
    names = sorted(os.listdir(data_dir), key=natural_key)
    return [os.path.join(data_dir, n) for n in names if os.path.isdir(os.path.join(data_dir, n))]


SIDE_CACHE = "test_data.cache"          # raw-array cache of test_data.pkl, written next to it (see _write_sidecar)
_MAGIC = 0x47454D43414348         # "GEMCACH"
_HDR = 8                          # int64 words: magic, n_frames, H, W, J, n_joint_coords (J*3), size and mtime_ns of the pickle


def _pickle_stamp(path):
    st = os.stat(os.path.join(path, "test_data.pkl"))
    return int(st.st_size), int(st.st_mtime_ns)


def _sidecar_layout(n, H, W, J):
    """byte offsets of (est_local f64 [n,J,3], gt f64 [n,J,3], cams f64 [n,4,4], heat f32 [n,H,W,J]) and the file size."""
    o_est = _HDR * 8
    o_gt = o_est + n * J * 3 * 8
    o_cam = o_gt + n * J * 3 * 8
    o_heat = (o_cam + n * 16 * 8 + 4095) // 4096 * 4096          # page-aligned: one aligned bulk read
    return o_est, o_gt, o_cam, o_heat, o_heat + n * H * W * J * 4


def _read_sidecar(path, small=True):
    """(header (n,H,W,J), small arrays or None, cache file, heat offset) when a valid cache of `<chunk>/test_data.pkl` exists,
    else None.  Valid = right magic, made from a pickle of EXACTLY this size and modification time (the header records both:
    a pickle replaced by `cp -p` / `rsync -t` / a restore, i.e. with an older time stamp, invalidates the cache too), exactly
    as long as its header says."""
    cache = os.path.join(path, SIDE_CACHE)
    try:
        with open(cache, "rb", buffering=0) as f:
            hdr = np.frombuffer(f.read(_HDR * 8), dtype=np.int64)
            if hdr.shape[0] != _HDR or hdr[0] != _MAGIC or (int(hdr[6]), int(hdr[7])) != _pickle_stamp(path):
                return None
            n, H, W, J = (int(v) for v in hdr[1:5])
            o_est, o_gt, o_cam, o_heat, total = _sidecar_layout(n, H, W, J)
            if os.fstat(f.fileno()).st_size != total:
                return None
            arrays = None
            if small:
                blob = np.frombuffer(f.read(o_cam + n * 16 * 8 - o_est), dtype=np.float64)
                arrays = {"est_local": blob[:n * J * 3].reshape(n, J, 3).copy(), "gt": blob[n * J * 3:2 * n * J * 3].reshape(n, J, 3).copy(),
                          "cams": blob[2 * n * J * 3:].reshape(n, 4, 4).copy()}
        return (n, H, W, J), arrays, cache, o_heat
    except (OSError, ValueError):
        return None


def _write_sidecar(path, c, heat):
    """One-time cache next to the pickle: ONE raw file holding the small arrays (float64) and the heat-maps as float32
    [N,H,W,J] -- what the device wants, read back with a single readinto of a pinned buffer instead of un-pickling N Python
    objects under the GIL.  Written atomically; a read-only data directory is simply left without a cache."""
    cache = os.path.join(path, SIDE_CACHE)
    try:
        h = np.ascontiguousarray(np.asarray(heat, dtype=np.float32))
        if h.ndim != 4 or c["est_local"].shape != (h.shape[0], h.shape[3], 3) or c["cams"].shape != (h.shape[0], 4, 4):
            return
        n, H, W, J = h.shape
        o_est, o_gt, o_cam, o_heat, total = _sidecar_layout(n, H, W, J)
        tmp = cache + ".tmp%d" % os.getpid()
        with open(tmp, "wb") as f:
            f.write(np.array([_MAGIC, n, H, W, J, J * 3, *_pickle_stamp(path)], dtype=np.int64).tobytes())
            for k in ("est_local", "gt", "cams"):
                f.write(np.ascontiguousarray(c[k], dtype=np.float64).tobytes())
            f.write(b"\0" * (o_heat - f.tell()))
            f.write(memoryview(h).cast("B"))
        os.replace(tmp, cache)
    except OSError:
        try:
            os.remove(cache)
        except OSError:
            pass


def _stage_to_device(fill, shape, device, dest=None):
    """fill(view) writes the float32 [shape] data into this reader thread's pinned staging buffer; one async H2D copy on the
    thread's own stream follows, into `dest` (a slice of the consumer's frame buffer: no concatenation afterwards) or into a
    fresh tensor.  Returns (device tensor, event)."""
    tl = _reader_local
    if getattr(tl, "stream", None) is None or tl.device != device:
        tl.stream, tl.device, tl.stage, tl.copied, tl.turn = torch.cuda.Stream(device=device), device, [None, None], [None, None], 0
    numel = int(np.prod(shape))
    # two staging buffers per reader: the next chunk is read from the page cache while the previous one is still on its way
    # over PCIe (with one buffer the copy engine idles for a whole read, 2.4 ms per chunk and reader)
    k = tl.turn
    tl.turn ^= 1
    if tl.stage[k] is None or tl.stage[k].numel() < numel:
        tl.stage[k] = torch.empty(max(numel, 1), dtype=torch.float32).pin_memory()
    if tl.copied[k] is not None:
        tl.copied[k].synchronize()                     # the copy that last used this staging buffer has left it
    stage = tl.stage[k]
    view = stage[:numel].view(shape).numpy()
    if numel:
        fill(view)
    with torch.cuda.stream(tl.stream):
        if dest is not None and tuple(dest.shape) == tuple(shape):
            t = dest.copy_(stage[:numel].view(shape), non_blocking=True)
        else:
            t = stage[:numel].view(shape).to(device, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(tl.stream)
    tl.copied[k] = ev
    return t, ev


def peek_frames(path):
    """Number of frames of a chunk without reading its arrays (from the cache's header), or None when unknown."""
    side = _read_sidecar(path, small=False)
    return None if side is None else side[0][0]


_SKIP_MIN = 128 * 1024            # bytes: payloads at least this long are not copied while un-pickling (pickle frames stay below it)


class _SkippingReader:
    """File object for `pickle.Unpickler` that SKIPS the large byte payloads (the raw data of the heat-map arrays): a skipped
    `readinto` costs no copy -- the buffer the unpickler allocated is only tagged with the payload's index, the position moves
    on -- and the payload's file offset is recorded.  The chunk's 100 heat-maps (24.6 MB) then never pass through Python objects:
    the reader thread brings the FILE as it is into pinned memory (one read, outside the GIL) and on to the device, where one
    strided copy picks the arrays out (`_stage_file_to_device`), instead of 100 bytes objects built under the GIL (1.5-3 ms per
    chunk whatever the number of reader threads).  Everything else (opcodes, frames, small arrays) is read by the memory map's
    own C methods, handed to the unpickler as they are: no Python frame per opcode and -- unlike a buffered file -- no system
    call, so a reader thread keeps the GIL for the ~0.5 ms it parses a chunk instead of handing it over at every read.  No
    `peek`, so the unpickler asks for exactly the bytes it needs and a payload always arrives as ONE request."""

    def __init__(self, mm):
        self._mm, self.segments = mm, []
        self.read, self.readline = mm.read, mm.readline        # (bound C methods)
        self._size = len(mm)

    def readinto(self, b):
        n = len(b)
        mm = self._mm
        if n < _SKIP_MIN:
            data = mm.read(n)
            b[:len(data)] = data
            return len(data)
        p = mm.tell()
        if p + n > self._size:
            return 0                                           # truncated file: the unpickler raises, the caller falls back
        mm.seek(n, 1)
        self.segments.append((p, n))
        b[:8] = _TAG.pack(len(self.segments) - 1)
        return n


_TAG = __import__("struct").Struct("<q")


def _load_pickle_skipping(path):
    """(dict of the small arrays, heat shape, [file offsets of the heat-maps], opened file) or None when the pickle is not of the
    expected form (then the caller un-pickles it the plain way): every heat-map must be a C-contiguous float32 array of one
    shape whose data the reader skipped."""
    import mmap
    f = open(os.path.join(path, "test_data.pkl"), "rb", buffering=0)
    try:
        mm = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ)
    except (ValueError, OSError):
        f.close()
        return None
    rd = _SkippingReader(mm)
    try:
        d = pickle.Unpickler(rd).load()
        heat = d["heatmap_list"] if isinstance(d, dict) else None
        if not isinstance(heat, (list, tuple)) or not heat or not rd.segments:
            return None
        shp, offs, seen = None, [], set()
        for a in heat:
            if not isinstance(a, np.ndarray) or a.dtype != np.float32 or not a.flags.c_contiguous or a.nbytes < _SKIP_MIN:
                return None
            k = _TAG.unpack(a.reshape(-1)[:2].tobytes())[0]
            if not (0 <= k < len(rd.segments)) or rd.segments[k][1] != a.nbytes or k in seen:
                return None
            if shp is None:
                shp = a.shape
            elif a.shape != shp:
                return None
            seen.add(k)
            offs.append(rd.segments[k][0])
        if len(seen) != len(rd.segments):               # some other large object lost its bytes: not a pickle for this path
            return None
        small = {k: v for k, v in d.items() if k != "heatmap_list"}
        ok = (f, (len(heat),) + tuple(shp), offs, small)
        f = None
        return ok
    except (pickle.UnpicklingError, EOFError, ValueError, TypeError, IndexError, AttributeError, ImportError, KeyError):
        return None
    finally:
        rd.read = rd.readline = rd._mm = None                   # (the bound methods keep the map exported)
        try:
            mm.close()
        except (BufferError, ValueError):
            pass
        if f is not None:
            f.close()


def _stage_file_to_device(f, shape, offs, device, dest=None):
    """The heat-maps of a chunk whose pickle holds them as `shape[0]` byte payloads of one length at the file offsets `offs`:
    the whole file -> this reader thread's pinned staging buffer (one read, outside the GIL) -> the device (one async copy on the
    thread's stream), then one strided byte copy on the device gathers the payloads -- equally spaced, as a pickler writes a list
    of equal arrays -- into `dest` (or a fresh tensor).  Payloads that are not equally spaced are scattered on the host instead
    (one os.preadv into the float staging buffer).  Returns (device tensor, event)."""
    n, nb = shape[0], int(np.prod(shape[1:])) * 4
    stride = offs[1] - offs[0] if n > 1 else nb
    if n > 1 and (stride < nb or any(offs[i + 1] - offs[i] != stride for i in range(n - 1))):
        fd = f.fileno()

        def fill(view):
            iov, want, scrap = [], 0, bytearray(65536)
            for i, o in enumerate(offs):
                gap = o - (offs[i - 1] + nb) if i else 0
                if gap < 0 or gap > len(scrap) or len(iov) > 1000:
                    iov = None
                    break
                if gap:
                    iov.append(memoryview(scrap)[:gap])
                iov.append(memoryview(view[i]).cast("B"))
                want += gap + nb
            if iov is not None and os.preadv(fd, iov, offs[0]) == want:
                return
            for i, o in enumerate(offs):
                if os.preadv(fd, [memoryview(view[i]).cast("B")], o) != nb:
                    raise IOError("short read of %s" % f.name)
        return _stage_to_device(fill, shape, device, dest)
    tl = _reader_local
    if getattr(tl, "stream", None) is None or tl.device != device:
        tl.stream, tl.device, tl.stage, tl.copied, tl.turn = torch.cuda.Stream(device=device), device, [None, None], [None, None], 0
    if getattr(tl, "fstage", None) is None or tl.fdevice != device:
        tl.fstage, tl.fimg, tl.fdevice = [None, None], [None, None], device
    first, span = offs[0], stride * (n - 1) + nb                  # the byte range of the file that holds the payloads
    k = tl.turn
    tl.turn ^= 1
    if tl.fstage[k] is None or tl.fstage[k].numel() < span:
        tl.fstage[k] = torch.empty(span + (1 << 20), dtype=torch.uint8).pin_memory()
        tl.fimg[k] = torch.empty(span + (1 << 20), dtype=torch.uint8, device=device)
    if tl.copied[k] is not None:
        tl.copied[k].synchronize()                     # the copy that last used this staging buffer has left it
    host = tl.fstage[k][:span]
    if os.preadv(f.fileno(), [memoryview(host.numpy())], first) != span:
        raise IOError("short read of %s" % f.name)
    with torch.cuda.stream(tl.stream):
        img = tl.fimg[k][:span]
        img.copy_(host, non_blocking=True)
        t = dest if (dest is not None and tuple(dest.shape) == tuple(shape) and dest.is_contiguous()) else \
            torch.empty(shape, dtype=torch.float32, device=device)
        t.view(torch.uint8).view(n, nb).copy_(img.as_strided((n, nb), (stride, 1)))
        ev = torch.cuda.Event()
        ev.record(tl.stream)
    tl.copied[k] = ev
    return t, ev


def load_chunk(path, device=None, sidecar=False, dest=None):
    """`<chunk>/test_data.pkl` (optimizer.py:315-324) as dense arrays; KeyError on a missing key like the reference.
    With `device`, the heat-maps (99 % of the bytes) go straight to that device from the calling thread.
    sidecar=True (opt-in: it writes a ~25 MB `test_data.cache` next to every pickle, a side effect the reference does not have):
    a raw-array cache of the pickle is used when present and written after the first un-pickling (the pickle
    holds 100 separate [64,64,15] arrays per chunk: un-pickling them is GIL-bound Python object work, 3 ms per chunk and
    thread; the cache is one 24.6 MB read straight into pinned memory, 70 GB/s over 8 reader threads)."""
    side = _read_sidecar(path) if sidecar else None
    if side is not None:
        (n, H, W, J), arrays, cache, o_heat = side
        shape = (n, H, W, J)
        c = {"path": path, **arrays}

        def fill(view):
            with open(cache, "rb", buffering=0) as f:
                f.seek(o_heat)
                got = f.readinto(memoryview(view).cast("B"))
            if got != view.nbytes:
                raise IOError("short read of %s" % cache)
        if device is None:
            c["heat"] = np.empty(shape, dtype=np.float32)
            if c["heat"].size:
                fill(c["heat"])
        else:
            c["heat"], c["heat_ready"] = _stage_to_device(fill, shape, device, dest)
        return c
    fast = _load_pickle_skipping(path) if (device is not None and not sidecar) else None
    if fast is not None:
        f, shape, offs, small = fast
        try:
            c = {"path": path,
                 "est_local": np.asarray(small["estimated_local_skeleton"], dtype=np.float64),
                 "gt": np.asarray(small["gt_global_skeleton"], dtype=np.float64),
                 "cams": np.asarray(small["camera_pose_list"], dtype=np.float64)}
            c["heat"], c["heat_ready"] = _stage_file_to_device(f, shape, offs, device, dest)
        finally:
            f.close()
        return c
    with open(os.path.join(path, "test_data.pkl"), "rb") as f:
        d = pickle.load(f)
    c = {"path": path,
         "est_local": np.asarray(d["estimated_local_skeleton"], dtype=np.float64),
         "gt": np.asarray(d["gt_global_skeleton"], dtype=np.float64),
         "cams": np.asarray(d["camera_pose_list"], dtype=np.float64)}
    heat = d["heatmap_list"]
    n = len(heat)
    shape = (n,) + tuple(np.shape(heat[0])) if n else (0, 64, 64, 15)
    if device is None:
        c["heat"] = np.asarray(heat, dtype=np.float32).reshape(shape)
    else:
        def fill(view):
            if isinstance(heat, (list, tuple)):
                np.stack(heat, out=view)
            else:
                np.copyto(view, np.asarray(heat, dtype=np.float32))
        c["heat"], c["heat_ready"] = _stage_to_device(fill, shape, device, dest)
    if sidecar and n:
        _write_sidecar(path, c, heat)
    return c


_pool, _pool_workers = None, 0


def _reader_pool(workers):
    """One pool of reader threads for the process (thread start-up costs milliseconds here; the threads also keep their
    pinned staging buffers and copy streams between calls)."""
    global _pool, _pool_workers
    if _pool is None or _pool_workers < workers:
        from concurrent.futures import ThreadPoolExecutor
        _pool, _pool_workers = ThreadPoolExecutor(max_workers=max(1, workers), thread_name_prefix="gem-reader"), workers
    return _pool


class ChunkStream:
    """Reads chunk pickles (or their raw-array caches) on `workers` background threads, at most `depth` chunks ahead of the
    consumer, and yields them in directory order.  The reads release the GIL (file -> pinned memory), so they overlap with
    each other and with the consumer's device work."""

    def __init__(self, paths, depth=8, workers=8, device=None, sidecar=False, dests=None):
        self._paths = list(paths)
        self._depth = max(1, depth)
        self._device = device
        self._sidecar = sidecar
        self._dests = dests or {}          # path -> device slice the chunk's heat-maps are copied into
        self._pool = _reader_pool(workers)

        self._pending, self._it = [], iter(self._paths)

    def prime(self):
        """Hand the first `depth` chunks to the readers now (otherwise that happens when the first chunk is asked for)."""
        while len(self._pending) < self._depth:
            p = next(self._it, None)
            if p is None:
                break
            self._pending.append(self._pool.submit(load_chunk, p, self._device, self._sidecar, self._dests.get(p)))
        return self

    def __iter__(self):
        pending = self._pending
        try:
            while True:
                self.prime()
                if not pending:
                    return
                yield pending.pop(0).result()          # a reader's exception (e.g. KeyError) surfaces here
        finally:
            for f in pending:
                f.cancel()
            for f in pending:                           # (running reads finish before the caller reuses their destinations)
                if not f.cancelled():
                    try:
                        f.result()
                    except Exception:
                        pass
            del pending[:]


_noise_pool = {}


def _draw_noise(rows, D):
    """One `torch.randn(rows[k], D)` per chunk from the global generator, in order (optimizer.py:261: `torch.randn_like` per
    stage call; D5) -- drawn straight into consecutive slices of ONE pinned buffer that is kept between calls: a fresh 0.4 MB
    tensor per chunk plus their concatenation cost more in first-touch page faults (5 ms per 240 windows) than in arithmetic,
    and the pinned block goes to the device in one asynchronous copy.  Returns the per-chunk views."""
    total = int(sum(rows))
    key = (D, torch.get_default_dtype())
    buf = _noise_pool.get(key)
    if buf is None or buf.shape[0] < total:
        buf = torch.empty(max(total, 1), D)
        if torch.cuda.is_available():
            buf = buf.pin_memory()
        _noise_pool[key] = buf
    out, r0 = [], 0
    for r in rows:
        v = buf[r0:r0 + r]
        if r:
            torch.randn(r, D, out=v)
        out.append(v)
        r0 += r
    return out


def _noise_block(views):
    """The concatenation of `_draw_noise`'s views without copying (they are consecutive slices of one buffer)."""
    if not views:
        return torch.empty(0, 0)
    total = sum(v.shape[0] for v in views)
    base = views[0]
    whole = base.as_strided((total, base.shape[1]), (base.shape[1], 1), base.storage_offset())
    return whole


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
                       device_metrics=True, verbose=True, seq_len=SEQ_LEN, overlap=OVERLAP, sidecar=False, timings=None):
    """Several sequences in ONE batched device call (BASELINE configs[2]: all test sequences concurrently on one GPU):
    the chunks of every directory of `data_dirs` go through the optimiser together, the reports are per sequence.
    Returns a list of (summary, per-chunk error dicts, estimated_pose, optimized_pose, gt_pose), one per directory, each
    exactly what `optimize_directory` returns; the noise is drawn sequence by sequence, chunk by chunk.
    sidecar=True keeps a raw-array cache of every pickle next to it (`load_chunk`; CLI: --cache true); the default reads the
    pickles only and leaves the data directory untouched.  The frame / noise buffers and reader threads this module keeps
    between calls (about 0.5 GB of HBM and pinned host memory per 2000-frame sequence) are shared by all calls of the process
    without locking -- one call at a time -- and are given back by `release_pools()`."""
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
    # When every chunk's frame count is known up front (cache metadata), the readers copy their heat-maps straight into
    # slices of ONE frame buffer per batch (two buffers alternate, so that the readers can fill the next batch while the
    # device still works on the current one): no concatenation, no 0.5 GB allocation per call.
    dests, batch_bufs = {}, []
    frames = [peek_frames(q) if sidecar else None for q in paths]
    if paths and all(f is not None for f in frames):
        per_batch = chunks_per_batch or len(paths)
        hs = optimizer.engine.heat_size if optimizer is not None else (64, 64)
        need = max(sum(frames[i:i + per_batch]) for i in range(0, len(paths), per_batch))
        pool = _heat_pool.setdefault(device, [])
        while len(pool) < (2 if chunks_per_batch and len(paths) > per_batch else 1):
            pool.append(None)
        for k in range(len(pool)):
            if pool[k] is None or pool[k].shape[0] < need or tuple(pool[k].shape[1:3]) != tuple(hs):
                pool[k] = torch.empty((need, hs[0], hs[1], 15), dtype=torch.float32, device=device)
        for bi, i in enumerate(range(0, len(paths), per_batch)):
            buf, f0 = pool[bi % len(pool)], 0
            batch_bufs.append((buf, sum(frames[i:i + per_batch])))
            for q, n in zip(paths[i:i + per_batch], frames[i:i + per_batch]):
                dests[q] = buf[f0:f0 + n]
                f0 += n
    # read-ahead: with the in-place frame buffers at most ONE batch ahead of the consumer (the buffer of batch k+2 is the
    # buffer of batch k: its readers may only start once batch k has left the device, i.e. when the consumer asks for more)
    depth = (chunks_per_batch or len(paths) or 1) if dests else max(16, (chunks_per_batch or 0) + 2)
    import time
    tick = [time.perf_counter()]

    def lap(name):          # developer timing (tools/whole_sequence_timing.py): wall time of the host phases
        if timings is not None:
            now = time.perf_counter()
            timings[name] = timings.get(name, 0.0) + (now - tick[0])
            tick[0] = now
    lap("plan")
    batch_iter = _batches(ChunkStream(paths, depth=depth, device=device, sidecar=sidecar, dests=dests).prime(), chunks_per_batch)
    known = paths and all(f is not None for f in frames) and opt is not None
    bi = -1
    while True:
        bi += 1
        eps = []
        if known:          # frame counts known up front: the noise is drawn while the readers work (same order, same generator: D5)
            per_batch = chunks_per_batch or len(paths)
            for n in frames[bi * per_batch:(bi + 1) * per_batch]:
                eps.append(2 * len(window_starts(n, seq_len, overlap)))
            eps = _draw_noise(eps, opt.engine.D)
            lap("noise")
        batch = next(batch_iter, None)
        if batch is None:
            break
        lap("wait_for_readers")
        starts, chunk_of, bounds, f_off = [], [], [], 0
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
        if not eps:
            eps = _draw_noise([2 * len(c["starts"]) for c in batch], opt.engine.D)
            lap("noise")
        if [e_.shape[0] for e_ in eps] != [2 * len(c["starts"]) for c in batch]:
            raise RuntimeError("chunk sizes changed between the cache headers and the data")
        w_local, w_global = opt.stage_weights(vae_weight, smoothness_weight, bone_length_weight, weight_3d, reproj_weight)
        for c in batch:
            torch.cuda.current_stream().wait_event(c["heat_ready"])
            c["heat"].record_stream(torch.cuda.current_stream())      # allocated on a reader's stream, consumed on this one
        if timings is not None:                  # developer timing only: separates the PCIe tail from the optimiser's own time
            torch.cuda.current_stream().synchronize()
            lap("h2d tail (timing runs only: synchronised)")
        if batch_bufs and all(c["heat"].data_ptr() == dests[c["path"]].data_ptr() for c in batch):
            heat_d = batch_bufs[bi][0][:batch_bufs[bi][1]]          # the readers filled the batch's frame buffer in place
        else:
            heat_d = batch[0]["heat"] if len(batch) == 1 else torch.cat([c["heat"] for c in batch])
        est_cat = np.concatenate([c["est_local"] for c in batch])
        cams_cat = np.concatenate([c["cams"] for c in batch])
        counts = [len(c["starts"]) for c in batch]
        equal = bool(device_metrics and counts and min(counts) == max(counts) and counts[0] > 0)
        prep = {}

        def prepare_reports():
            # equal chunks (the reference's 100-frame chunks): the sequences main() returns besides the optimised one are built
            # for ALL windows of the batch at once, the overlap merges vectorised over the chunks.  What does not depend on the
            # optimiser's result is computed while the device works.
            if not equal:
                return
            gt_cat = np.concatenate([c["gt"] for c in batch])
            idx = np.concatenate(starts)[:, None] + np.arange(seq_len)[None]
            prep["cam_w"] = cams_cat[idx]
            prep["est_m"] = merge_chunks(to_global_numpy(relative_global_numpy(est_cat[idx], prep["cam_w"]), prep["cam_w"]), len(batch), overlap)
            prep["gt_m"] = merge_chunks(gt_cat[idx], len(batch), overlap)
        mid_local, opt_global, _ = opt.run(est_cat, cams_cat, heat_d, np.concatenate(starts), np.concatenate(chunk_of), bounds, w_local,
                                           w_global, eps=_noise_block(eps), keep_device=True, timings=timings,
                                           while_device_runs=prepare_reports)
        lap("optimise (enqueue + device + read-back of the stats)")
        mid_np = mid_local.cpu().numpy()
        if equal:
            # ... the error reports of all chunks are enqueued back to back and read back with ONE synchronisation
            e, nb, wpc = opt.engine, len(batch), counts[0]
            cam_w, est_m, gt_m = prep["cam_w"], prep["est_m"], prep["gt_m"]
            mid_m = merge_chunks(to_global_numpy(relative_global_numpy(mid_np, cam_w), cam_w), nb, overlap)
            fpc = est_m.shape[1]
            opt_d = e.merge_windows(opt_global, nb, overlap=overlap, smooth=bool(final_smooth))          # [nb*fpc,15,3] f64, device
            est_d, mid_d, gt_d = (torch.as_tensor(x.reshape(nb * fpc, 15, 3), device=device) for x in (est_m, mid_m, gt_m))
            reps = torch.stack([e.calculate_errors_device(est_d[k * fpc:(k + 1) * fpc], mid_d[k * fpc:(k + 1) * fpc],
                                                          opt_d[k * fpc:(k + 1) * fpc], gt_d[k * fpc:(k + 1) * fpc]) for k in range(nb)])
            reps, opt_m = reps.cpu().numpy(), opt_d.cpu().numpy().reshape(nb, fpc, 15, 3)
            for k, c in enumerate(batch):
                res = OrderedDict((key, float(reps[k, i])) for i, key in enumerate(e.ERROR_KEYS))
                res["joints_error"] = reps[k, 17:].copy()
                gi = group_of[c["path"]]
                results[gi].append(res)
                est_all[gi].extend(list(est_m[k])); opt_all[gi].extend(list(opt_m[k])); gt_all[gi].extend(list(gt_m[k]))
                if verbose and res["bone_length_aligned_optimized_mpjpe"] > res["bone_length_aligned_mid_optimized_mpjpe"]:
                    print(res)
            lap("sequences + reports")
            continue
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


def release_pools():
    """Give back what this module keeps between calls: the per-device frame buffers the readers fill, the pinned noise blocks
    and the reader threads (with their pinned staging buffers).  Not to be called while another call is in flight."""
    global _pool, _pool_workers
    _heat_pool.clear()
    _noise_pool.clear()
    pool, _pool, _pool_workers = _pool, None, 0
    if pool is not None:
        pool.shutdown(wait=True)
    if torch.cuda.is_available():
        torch.cuda.empty_cache()


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
    p.add_argument("--cache", default=False, type=truthy, help="keep a raw-array cache (test_data.cache, ~25 MB) next to every pickle: "
                   "4x faster reads from the second run on")
    a = p.parse_args()
    if a.save:
        raise NotImplementedError("--save writes open3d meshes (optimizer.py:452-504): outside the hot path")
    optimize_directory(a.data_path, a.camera, a.vae, a.gmm, a.smooth, a.bone_length, a.weight_3d, a.reproj_weight,
                       final_smooth=a.final_smooth, merge=a.merge, chunks_per_batch=a.chunks_per_batch, sidecar=a.cache)


if __name__ == "__main__":
    _cli()
