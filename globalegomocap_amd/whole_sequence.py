"""Whole-sequence driver: every chunk directory of a sequence in ONE batched device call.

The reference's `optimize_whole_sequence.py` walks the chunk directories of a sequence (natsorted), calls
`main()` on each -- 12 windows, one after the other -- and averages the per-chunk error dicts
(`optimize_whole_sequence.py:48-118`).  Here the library reads the pickles itself (`parse_chunk` / `read_file` /
`gather_heat`: the heat-maps go file -> pinned memory -> HBM without becoming Python objects, loadmat's Fortran order and
float64 are undone by a kernel), all windows of a batch of chunks go through the optimiser together (BASELINE configs[1]: a
2000-frame sequence = 20 chunks = 240 windows per call), batches are pipelined (the next one's files arrive while this one
computes), and the per-chunk merge / smoothing / error report run on the device as well.  Results, keys, printed summary and the order in which the reparameterisation noise is
drawn (chunk by chunk, window by window, local then global) follow the reference.

    python -m globalegomocap_amd.whole_sequence --data_path data/jian3
"""
import ctypes as C
import os
import pickle
import re
import threading
from collections import OrderedDict

import numpy as np
import torch

from . import _capi
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


def natural_key(name):
    """Sort key equivalent to natsort.natsorted (default algorithm: case-sensitive text, unsigned integers) for directory
    names like chunk_2 < chunk_10 (optimize_whole_sequence.py:48)."""
    return [int(t) if t.isdigit() else t for t in re.split(r"(\d+)", name)]


def list_chunks(data_dir):
    """Chunk directories in the reference's order (`natsorted(os.listdir(data_dir))`, directories only)."""
    names = sorted(os.listdir(data_dir), key=natural_key)
    return [os.path.join(data_dir, n) for n in names if os.path.isdir(os.path.join(data_dir, n))]


KEYS = ("estimated_local_skeleton", "gt_global_skeleton", "camera_pose_list", "heatmap_list")     # read at optimizer.py:318-324, in this order
_K_EST, _K_GT, _K_CAM, _K_HEAT = range(4)
SLICE_BYTES = int(os.environ.get("GEM_WS_SLICE_MB", 8)) << 20            # a file crosses PCIe in slices of this size: the next one is read while the last one is on its way
_PAD = 4096                      # file images are laid out on 4 KB boundaries, with at least 8 bytes of slack behind each

_reader_local = threading.local()      # per reader thread: two pinned staging buffers, (load_chunk only) a device image of the file
_heat_pool = {}                        # device -> per batch in flight: [frame buffer, file-image arena] (kept between calls)
_C_KEYS = None


class ParsedChunk(dict):
    """What `parse_chunk` found in `<chunk>/test_data.pkl`: "path", "est_local" / "gt" / "cams" (dense float64 arrays), "n" frames,
    "heat_shape" (H, W, J), and EITHER "heat_offsets" (int64 [n]: where every heat-map's raw data lie in the file) with
    "heat_dtype" / "heat_fortran" -- the library located them, they reach the device without passing through Python (`read_file` +
    `gather_heat`) -- OR "heat_list", the un-pickled list (files outside the library reader's subset)."""


def parse_chunk(path, native=True):
    """`<chunk>/test_data.pkl` (optimizer.py:315-324) without its heat-maps' data: the small arrays dense, the heat-maps located.
    The library interprets the pickle itself (gem_chunk_open: no Python object per array, no GIL while it runs, nothing the file
    names is imported or called); a file outside its subset -- protocol 2, an entry that is no list of equally shaped float
    arrays -- is un-pickled the ordinary way instead.  KeyError on a missing key, like the reference."""
    global _C_KEYS
    file = os.path.join(path, "test_data.pkl")
    if native:
        lib = _capi.load_library()
        if _C_KEYS is None:
            _C_KEYS = (C.c_char_p * len(KEYS))(*[k.encode() for k in KEYS])
        h = C.c_void_p()
        if lib.gem_chunk_open(os.fsencode(file), _C_KEYS, len(KEYS), C.byref(h)) == 0:
            try:
                info = (C.c_int64 * 8)()
                c, ok = ParsedChunk(path=path), True
                for k, name in ((_K_EST, "est_local"), (_K_GT, "gt"), (_K_CAM, "cams"), (_K_HEAT, None)):
                    _capi.check(lib.gem_chunk_info(h, k, info), lib)
                    n, ndim = int(info[0]), int(info[1])
                    if n < 0:
                        raise KeyError(KEYS[k])
                    if ndim < 0 or (name is None and (n == 0 or ndim != 3)):
                        ok = False                    # ragged, or heat-maps that are not [H,W,J] arrays: the ordinary way
                        break
                    if name is not None:
                        c[name] = np.empty((n,) + tuple(info[4:4 + ndim]), dtype=np.float64)
                        if n:
                            _capi.check(lib.gem_chunk_gather_f64(h, k, c[name].ctypes.data, c[name].size), lib)
                    else:
                        offs = np.empty(n, dtype=np.int64)
                        _capi.check(lib.gem_chunk_offsets(h, k, offs.ctypes.data, n), lib)
                        c.update(n=n, heat_shape=tuple(int(v) for v in info[4:7]), heat_dtype=int(info[2]), heat_fortran=int(info[3]),
                                 heat_offsets=offs, file_bytes=int(lib.gem_chunk_bytes(h)))
                if ok:
                    return c
            finally:
                lib.gem_chunk_close(h)
    with open(file, "rb") as f:
        d = pickle.load(f)
    c = ParsedChunk(path=path,
                    est_local=np.asarray(d["estimated_local_skeleton"], dtype=np.float64),
                    gt=np.asarray(d["gt_global_skeleton"], dtype=np.float64),
                    cams=np.asarray(d["camera_pose_list"], dtype=np.float64))
    heat = d["heatmap_list"]
    c["heat_list"], c["n"] = heat, len(heat)
    c["heat_shape"] = tuple(np.shape(heat[0])) if len(heat) else (64, 64, 15)
    return c


N_COPY_STREAMS = int(os.environ.get("GEM_WS_STREAMS", 2))
STREAM_PRIORITY = int(os.environ.get("GEM_WS_PRIORITY", 0))          # (0 = torch's default pool; -1 measured the same: tools/r06_stream_prio.sh)
_copy_streams = {}
_report_streams = {}
_copy_lock = threading.Lock()


def copy_stream(device):
    """The stream this reader thread copies on: the threads share N_COPY_STREAMS streams per device (handed out round-robin).
    Few, not one per thread: the runtime maps streams onto a handful of hardware queues, and a compute stream that lands in
    the same queue as a copying stream has its kernels held up behind that stream's copies (measured: a 7.6 ms optimiser call
    took 13.6 ms beside eight copying streams).  More than one, because copies of one stream run strictly one after the
    other with a gap between them."""
    with _copy_lock:
        st = _copy_streams.setdefault(device, [[], 0])
        if len(st[0]) < N_COPY_STREAMS:
            st[0].append(torch.cuda.Stream(device=device, priority=STREAM_PRIORITY))
        st[1] += 1
        return st[0][(st[1] - 1) % len(st[0])]


def _thread_state(device):
    tl = _reader_local
    if getattr(tl, "stream", None) is None or tl.device != device:
        tl.stream, tl.device = copy_stream(device), device
        tl.stage, tl.copied, tl.turn, tl.image = [None, None], [None, None], 0, None
    return tl


def _staging(tl, nbytes):
    """One of this reader thread's two pinned staging buffers (uint8, at least nbytes), free to be written: the copy that last
    used it has left it.  Two alternate, so that the next file is read while the last one is still crossing PCIe."""
    k = tl.turn
    tl.turn ^= 1
    if tl.stage[k] is None or tl.stage[k].numel() < nbytes:
        tl.stage[k] = torch.empty(nbytes + (1 << 20), dtype=torch.uint8).pin_memory()
    if tl.copied[k] is not None:
        tl.copied[k].synchronize()
    return k, tl.stage[k]


def read_file(file, device, image, size=None):
    """`file` -> this reader thread's pinned staging buffer -> `image` (uint8 device tensor of at least the file's size + 8), in
    slices, on the thread's own stream (gem_file_stage; the GIL is free for the whole call).  Returns (event, file size): the
    image is complete once the event has passed."""
    lib = _capi.load_library()
    tl = _thread_state(device)
    size = os.path.getsize(file) if size is None else size
    k, stage = _staging(tl, size + 8)
    got = C.c_int64()
    rc = lib.gem_file_stage(os.fsencode(file), device.index if device.index is not None else torch.cuda.current_device(),
                            C.c_void_p(stage.data_ptr()), C.c_void_p(image.data_ptr()), min(stage.numel(), image.numel()), SLICE_BYTES,
                            C.byref(got), C.c_void_p(tl.stream.cuda_stream))
    ev = torch.cuda.Event()
    ev.record(tl.stream)
    tl.copied[k] = ev
    _capi.check(rc, lib)
    return ev, got.value


def gather_heat(c, image, offsets_d, dest, stream=None):
    """The heat-maps of parsed chunk `c` out of the device image of its file: ONE kernel (gem_heat_gather) on the current stream
    picks the arrays out at `offsets_d` (int64 device tensor [n]), undoes loadmat's Fortran order and rounds float64 to float32
    -> dest [n,H,W,J] float32."""
    lib = _capi.load_library()
    H, W, J = c["heat_shape"]
    st = stream if stream is not None else torch.cuda.current_stream()
    _capi.check(lib.gem_heat_gather(C.c_void_p(image.data_ptr()), c["file_bytes"], C.c_void_p(offsets_d.data_ptr()), c["n"], H, W, J,
                                    c["heat_dtype"], c["heat_fortran"], C.c_void_p(dest.data_ptr()), C.c_void_p(st.cuda_stream)), lib)


def stage_list(c, device, dest=None):
    """The ordinary way for files the library's reader declined: the un-pickled heat-map list is stacked into this thread's pinned
    staging buffer and copied to `dest` (or a fresh tensor) on the thread's stream.  Returns (tensor, event)."""
    shape = (c["n"],) + tuple(c["heat_shape"])
    tl = _thread_state(device)
    numel = int(np.prod(shape))
    heat = c.pop("heat_list")
    k, stage = _staging(tl, numel * 4)
    view = stage[:numel * 4].view(torch.float32).view(shape)
    if numel:
        if isinstance(heat, (list, tuple)):
            np.stack(heat, out=view.numpy(), casting="same_kind")
        else:
            np.copyto(view.numpy(), np.asarray(heat), casting="same_kind")
    with torch.cuda.stream(tl.stream):
        if dest is not None and tuple(dest.shape) == tuple(shape) and dest.dtype == torch.float32:
            t = dest.copy_(view, non_blocking=True)
        else:
            t = view.to(device, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(tl.stream)
    tl.copied[k] = ev
    return t, ev


def load_chunk(path, device=None, dest=None):
    """`<chunk>/test_data.pkl` as dense arrays: "est_local", "gt", "cams" float64, "heat" float32 [n,H,W,J] -- a numpy array, or
    with `device` a tensor there (`dest` when it has that shape) plus "heat_ready", the event after which it may be read; the
    copies run from the calling thread, on a stream of its own."""
    if device is None:
        c = parse_chunk(path, native=False)
        heat = c.pop("heat_list")
        c["heat"] = np.asarray(heat, dtype=np.float32).reshape((c["n"],) + tuple(c["heat_shape"]))
        return c
    c = parse_chunk(path)
    if "heat_offsets" not in c:
        c["heat"], c["heat_ready"] = stage_list(c, device, dest)
        return c
    tl = _thread_state(device)
    shape = (c["n"],) + tuple(c["heat_shape"])
    with torch.cuda.stream(tl.stream):
        if tl.image is None or tl.image.numel() < c["file_bytes"] + 8:
            tl.image = torch.empty(c["file_bytes"] + (1 << 20), dtype=torch.uint8, device=device)
        read_file(os.path.join(path, "test_data.pkl"), device, tl.image, c["file_bytes"])
        t = dest if (dest is not None and tuple(dest.shape) == shape and dest.is_contiguous() and dest.dtype == torch.float32) else \
            torch.empty(shape, dtype=torch.float32, device=device)
        gather_heat(c, tl.image, torch.from_numpy(c["heat_offsets"]).to(device), t, tl.stream)
        ev = torch.cuda.Event()
        ev.record(tl.stream)
    c["heat"], c["heat_ready"] = t, ev
    return c


_pools = {}


def cpus_near(device):
    """The CPUs of the NUMA node the device hangs off (its PCIe root), as far as this process may run on them -- or None when
    the platform does not say.  The readers copy page cache -> pinned memory (which the runtime places next to the device):
    from the other socket that copy crosses the inter-socket links and the read + host-to-device pipeline of a 2000-frame
    sequence took 14.5 instead of 10.5 ms (tools/r06_numa_probe.py)."""
    try:
        p = torch.cuda.get_device_properties(device)
        bdf = "%04x:%02x:%02x.0" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
        node = int(open("/sys/bus/pci/devices/%s/numa_node" % bdf).read())
        if node < 0:
            return None
        cpus = set()
        for part in open("/sys/devices/system/node/node%d/cpulist" % node).read().strip().split(","):
            lo, _, hi = part.partition("-")
            cpus.update(range(int(lo), int(hi or lo) + 1))
        cpus &= os.sched_getaffinity(0)
        return cpus or None
    except (OSError, ValueError, AttributeError, RuntimeError):
        return None


def _pool(name, workers, cpus=None):
    """The process's pools of reader threads (thread start-up costs milliseconds here; the readers also keep their pinned
    staging buffers between calls).  `cpus`: the threads of a NEW pool are confined to these."""
    p = _pools.get(name)
    if p is None or p[1] < workers:
        from concurrent.futures import ThreadPoolExecutor

        def confine():
            if cpus:
                try:
                    os.sched_setaffinity(0, cpus)
                except OSError:
                    pass
        p = _pools[name] = (ThreadPoolExecutor(max_workers=max(1, workers), thread_name_prefix="gem-" + name, initializer=confine), workers)
    return p[0]


class ChunkStream:
    """Chunks in directory order, read `depth` ahead of the consumer by the reader threads: parsed (and with `device`, their
    heat-maps on their way to it) when they are handed out."""

    def __init__(self, paths, depth=8, workers=8, device=None):
        self._paths, self._depth, self._device = list(paths), max(1, depth), device
        self._pool = _pool("read", workers)
        self._pending, self._it = [], iter(self._paths)

    def prime(self):
        while len(self._pending) < self._depth:
            p = next(self._it, None)
            if p is None:
                break
            self._pending.append(self._pool.submit(load_chunk, p, self._device))
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
            _drain(pending)


def _drain(futures):
    """Cancel what has not started, wait for what has (its buffers are about to be reused or released)."""
    futures[:] = [f for f in futures if not f.done()]          # (the normal way out: everything has long finished)
    for f in futures:
        f.cancel()
    for f in futures:
        if not f.cancelled():
            try:
                f.result()
            except Exception:
                pass
    del futures[:]


_noise_pool = {}
N_BUFFERS = 3                     # batches in flight: one computing, one arriving, one being reported


def _draw_noise(rows, D, slot=0):
    """One `torch.randn(rows[k], D)` per chunk from the global generator, in order (optimizer.py:261: `torch.randn_like` per
    stage call; D5) -- drawn straight into consecutive slices of ONE pinned buffer that is kept between calls (one per batch in
    flight, `slot`): a fresh 0.4 MB tensor per chunk plus their concatenation cost more in first-touch page faults (5 ms per
    240 windows) than in arithmetic, and the pinned block goes to the device in one asynchronous copy.  Returns the block
    [sum(rows), D]."""
    total = int(sum(rows))
    key = (D, torch.get_default_dtype(), slot)
    buf = _noise_pool.get(key)
    if buf is None or buf.shape[0] < total:
        buf = torch.empty(max(total, 1), D)
        if torch.cuda.is_available():
            buf = buf.pin_memory()
        _noise_pool[key] = buf
    r0 = 0
    for r in rows:
        if r:
            torch.randn(r, D, out=buf[r0:r0 + r])
        r0 += r
    return buf[:total]


class _Scratch:
    """A pinned block per batch in flight for the SMALL arrays (poses, cameras, window tables, payload offsets, report inputs):
    they go to the device with asynchronous copies from it.  (A pageable copy blocks the calling thread until the copy engines
    get to it -- behind the readers' 4 MB slices that is milliseconds per array.)"""

    def __init__(self, device):
        self.device, self.buf, self.at = device, None, 0

    def reset(self, nbytes):
        if self.buf is None or self.buf.numel() < nbytes:
            self.buf = torch.empty(2 * nbytes, dtype=torch.uint8).pin_memory()
        self.at = 0

    def upload(self, a, dtype):
        a = np.asarray(a)
        n = int(a.size) * torch.empty(0, dtype=dtype).element_size()
        if self.buf is None or self.at + n + 64 > self.buf.numel():          # (more than reset() was told: an ordinary copy)
            return torch.as_tensor(a, dtype=dtype).to(self.device).contiguous()
        view = self.buf[self.at:self.at + n].view(dtype).view(a.shape)
        self.at += (n + 63) // 64 * 64
        np.copyto(view.numpy(), a, casting="unsafe")
        return view.to(self.device, non_blocking=True)


class _Batch:
    """One device call's worth of chunks on its way through the pipeline (see optimize_sequences)."""
    __slots__ = ("index", "paths", "files", "sizes", "parsing", "chunks", "reading", "images", "noise", "pending", "starts", "bounds",
                 "counts", "est_cat", "cams_cat", "prep", "heat", "dests", "listed", "offsets", "report", "weights", "done")

    def __init__(self, index, paths):
        self.index, self.paths = index, paths
        self.files = [os.path.join(q, "test_data.pkl") for q in paths]
        self.sizes = self.parsing = self.chunks = self.reading = self.images = self.noise = self.pending = self.report = None


def optimize_sequences(data_dirs, camera_model_path, vae_weight=0.0, gmm_weight=0.0, smoothness_weight=0.001,
                       bone_length_weight=0.01, weight_3d=0.01, reproj_weight=0.01, final_smooth=True, merge=True,
                       global_vae_path=GLOBAL_VAE_PATH, local_vae_path=LOCAL_VAE_PATH, chunks_per_batch=None, optimizer=None,
                       device_metrics=True, verbose=True, seq_len=SEQ_LEN, overlap=OVERLAP, timings=None, per_sequence=False):
    """Several sequences through the device: the chunks of every directory of `data_dirs`, `chunks_per_batch` per device call
    (default: all of them in ONE call -- BASELINE configs[2]: all test sequences concurrently on one GPU; per_sequence=True: one
    call per directory), the reports per sequence.  Returns a list of (summary, per-chunk error dicts, estimated_pose,
    optimized_pose, gt_pose), one per directory, each exactly what `optimize_directory` returns; the noise is drawn sequence by
    sequence, chunk by chunk.

    The files cross PCIe as they are (`read_file`: file -> pinned memory -> an image of the file in HBM, on 8 reader threads,
    nothing of it passing through Python) WHILE the library interprets the pickles on other threads (`parse_chunk`), the noise
    is drawn on a thread of its own (same generator, same order: D5) and the main thread uploads the small arrays; one kernel
    per chunk then picks the heat-maps out of the images (`gather_heat`).  Batches are pipelined: batch k+1's files are read
    behind batch k's, so they arrive while batch k is on the device, and batch k+1 is enqueued before batch k's report is read
    back.  The frame buffers, file-image arenas, noise blocks and reader threads this module keeps between calls (about 1 GB
    of HBM and 0.4 GB of pinned host memory per 2000-frame batch in flight, at most three) are shared by all calls of the
    process without locking -- one call at a time -- and are given back by `release_pools()`."""
    del gmm_weight, merge                       # accepted and unused, as in the reference
    import time
    tick = [time.perf_counter()]
    t_begin = tick[0]

    def lap(name):          # developer timing (tools/whole_sequence_timing.py): wall time of the main thread's phases
        if timings is not None:
            now = time.perf_counter()
            timings[name] = timings.get(name, 0.0) + (now - tick[0])
            timings.setdefault("_log", []).append((round((now - t_begin) * 1e3, 2), name))
            tick[0] = now
    groups, group_of = [], {}
    for gi, d in enumerate(data_dirs):
        ps = list_chunks(d)
        if not ps:
            raise FileNotFoundError("no chunk directories under %s" % d)
        for q in ps:
            group_of[q] = gi
        groups.append(ps)
    if per_sequence:
        lists = [g[i:i + (chunks_per_batch or len(g))] for g in groups for i in range(0, len(g), chunks_per_batch or len(g))]
    else:
        flat = [q for g in groups for q in g]
        lists = [flat[i:i + (chunks_per_batch or len(flat))] for i in range(0, len(flat), chunks_per_batch or len(flat))]
    batches = [_Batch(i, l) for i, l in enumerate(lists)]
    n_groups = len(data_dirs)
    opt = [optimizer]
    results, est_all, opt_all, gt_all, raw_rows = ([[] for _ in range(n_groups)] for _ in range(5))
    device = torch.device("cuda", torch.cuda.current_device())
    parse_pool, read_pool, noise_pool = _pool("parse", 8), _pool("read", 8, cpus_near(device)), _pool("noise", 1)
    slots = _heat_pool.setdefault(device, [[None, None, _Scratch(device)] for _ in range(N_BUFFERS)])
    submitted = []                               # every future handed to a pool (drained on the way out, whatever happens)

    def start(b):
        """Batch b's files start moving: read tasks (they need the files' sizes only) and, beside them, the parse tasks."""
        if b.reading is not None:
            return
        b.sizes = [os.path.getsize(f) for f in b.files]          # (FileNotFoundError here, like the reference's open())
        at, total = [], 0
        for sz in b.sizes:
            at.append(total)
            total += (sz + 8 + _PAD - 1) // _PAD * _PAD
        slot = slots[b.index % N_BUFFERS]
        if slot[1] is None or slot[1].numel() < total:
            slot[1] = None
            slot[1] = torch.empty(total, dtype=torch.uint8, device=device)
        b.images = [slot[1][o:o + (sz + 8 + _PAD - 1) // _PAD * _PAD] for o, sz in zip(at, b.sizes)]
        b.reading = [read_pool.submit(read_file, f, device, img, sz) for f, img, sz in zip(b.files, b.images, b.sizes)]
        b.parsing = [parse_pool.submit(parse_chunk, q) for q in b.paths]
        submitted.extend(b.reading + b.parsing)

    def prepare(b):
        """Everything of batch b that needs neither its heat-maps nor the device's attention: parses awaited, window tables, the
        small arrays on their way to the device, the noise being drawn, the report's result-independent half."""
        start(b)
        b.chunks = [f.result() for f in b.parsing]                # (KeyError etc. surface here)
        lap("wait for the parses")
        starts, chunk_of, bounds, f_off = [], [], [], 0
        for ci, c in enumerate(b.chunks):
            if verbose:
                print("running data: {}".format(c["path"]))
            s = window_starts(len(c["est_local"]), seq_len, overlap)
            c["starts"] = s
            starts.append(s + f_off)
            chunk_of.append(np.full(len(s), ci, dtype=np.int64))
            bounds.append((f_off, f_off + len(c["est_local"])))
            f_off += len(c["est_local"])
        n_win = int(sum(len(s) for s in starts))
        if opt[0] is None:
            opt[0] = SequenceOptimizer(camera_model_path, global_vae_path, local_vae_path, max_windows=max(n_win, 1), seq_len=seq_len)
        if n_win > opt[0].engine.max_windows:
            raise ValueError("%d windows in one batch exceed the engine's max_windows=%d: pass chunks_per_batch" %
                             (n_win, opt[0].engine.max_windows))
        b.noise = noise_pool.submit(_draw_noise, [2 * len(c["starts"]) for c in b.chunks], opt[0].engine.D, b.index % N_BUFFERS)
        submitted.append(b.noise)
        b.listed = [(i, read_pool.submit(stage_list, c, device)) for i, c in enumerate(b.chunks) if "heat_offsets" not in c]
        submitted.extend(f for _, f in b.listed)      # (files the library's reader declined: stacked on the host)
        b.weights = opt[0].stage_weights(vae_weight, smoothness_weight, bone_length_weight, weight_3d, reproj_weight)
        b.est_cat = np.concatenate([c["est_local"] for c in b.chunks])
        b.cams_cat = np.concatenate([c["cams"] for c in b.chunks])
        b.starts, b.bounds, b.counts = starts, bounds, [len(c["starts"]) for c in b.chunks]
        lap("window tables")
        slot = slots[b.index % N_BUFFERS]
        slot[2].reset(len(b.est_cat) * (45 * 4 + 16 * 8 + 8 + 3 * 45 * 8) + 16 * n_win + 8192)
        b.prep = opt[0].prepare(b.est_cat, b.cams_cat, np.concatenate(starts), np.concatenate(chunk_of), bounds, timings=timings,
                                upload=slot[2].upload)
        # the batch's frame buffer (one shape of heat-map: the chunks' frames side by side) and the table of payload offsets
        frames = sum(c["n"] for c in b.chunks)
        shapes = {tuple(c["heat_shape"]) for c in b.chunks}
        b.heat, b.dests = None, [None] * len(b.chunks)
        if len(shapes) == 1 and frames:
            hs = next(iter(shapes))
            if slot[0] is None or slot[0].shape[0] < frames or tuple(slot[0].shape[1:]) != hs:
                slot[0] = None
                slot[0] = torch.empty((frames,) + hs, dtype=torch.float32, device=device)
            b.heat = slot[0][:frames]
            b.dests = [b.heat[lo:hi] for lo, hi in bounds]
        # (where every heat-map's raw data lie in the ARENA of file images: the file's place in the arena + the array's place in the file)
        native = [c["heat_offsets"] + (img.data_ptr() - slot[1].data_ptr()) for c, img in zip(b.chunks, b.images) if "heat_offsets" in c]
        b.offsets = slot[2].upload(np.concatenate(native), torch.int64) if native else None
        lap("small uploads")
        # the report's half that does not depend on the optimiser's result (equal chunks -- the reference's 100-frame chunks:
        # the sequences main() returns besides the optimised one, for ALL windows of the batch at once, the overlap merges
        # vectorised over the chunks), computed and uploaded while the files are still arriving
        counts = b.counts
        b.report = None
        if device_metrics and counts and min(counts) == max(counts) and counts[0] > 0:
            gt_cat = np.concatenate([c["gt"] for c in b.chunks])
            idx = np.concatenate(starts)[:, None] + np.arange(seq_len)[None]
            cam_w = b.cams_cat[idx]
            est_m = merge_chunks(to_global_numpy(relative_global_numpy(b.est_cat[idx], cam_w), cam_w), len(b.chunks), overlap)
            gt_m = merge_chunks(gt_cat[idx], len(b.chunks), overlap)
            # (stage one's global sequence: C0 (C0^-1 C_t) X as ONE transform per frame, composed here in the reference's order --
            # utils/utils.py:99-112 then optimizer.py:302-308 -- so that the result-dependent half is a multiply-add)
            A = np.matmul(cam_w[:, :1], np.matmul(np.linalg.inv(cam_w[:, 0])[:, None], cam_w))
            b.report = {"mid_A": np.ascontiguousarray(np.moveaxis(A[..., :3, :], (-2, -1), (0, 1))[..., None]), "est_m": est_m, "gt_m": gt_m,      # mid_A [3,4,W,T,1]
                        "est_d": slot[2].upload(est_m.reshape(-1, 15, 3), torch.float64),
                        "gt_d": slot[2].upload(gt_m.reshape(-1, 15, 3), torch.float64)}
        lap("report preparation")

    def fire(b):
        """Batch b goes to the device: its files' last copies awaited (issued, not finished), one gather kernel per chunk, the
        optimiser's call enqueued behind them.  Nothing waits for the device."""
        eps = b.noise.result()
        lap("wait for the noise")
        cur = torch.cuda.current_stream()
        at, parts = 0, [None] * len(b.chunks)
        native = [i for i, c in enumerate(b.chunks) if "heat_offsets" in c]
        for i in native:
            ev, _ = b.reading[i].result()             # (the file's last copy has been issued: its event is recorded)
            cur.wait_event(ev)
        kinds = {(b.chunks[i]["heat_dtype"], b.chunks[i]["heat_fortran"], tuple(b.chunks[i]["heat_shape"])) for i in native}
        if b.heat is not None and len(native) == len(b.chunks) and len(kinds) == 1 and sum(c["n"] for c in b.chunks) <= 65535:
            # every chunk's file image lies in ONE arena and their frames side by side in the batch's frame buffer: one launch picks all
            # heat-maps out (b.offsets was built relative to the arena in prepare())
            arena = slots[b.index % N_BUFFERS][1]
            c0 = b.chunks[0]
            whole = ParsedChunk(n=sum(c["n"] for c in b.chunks), heat_shape=c0["heat_shape"], heat_dtype=c0["heat_dtype"], heat_fortran=c0["heat_fortran"],
                                file_bytes=arena.numel())
            gather_heat(whole, arena, b.offsets, b.heat, cur)
            for i in native:
                parts[i] = b.dests[i]
        else:
            for i in native:
                c = b.chunks[i]
                parts[i] = b.dests[i] if b.dests[i] is not None else torch.empty((c["n"],) + tuple(c["heat_shape"]), dtype=torch.float32, device=device)
                base = b.images[i].data_ptr() - slots[b.index % N_BUFFERS][1].data_ptr()
                gather_heat(ParsedChunk(c, file_bytes=base + c["file_bytes"]), slots[b.index % N_BUFFERS][1], b.offsets[at:at + c["n"]], parts[i], cur)
                at += c["n"]
        for i, f in b.listed:
            t, ev = f.result()
            cur.wait_event(ev)
            t.record_stream(cur)
            parts[i] = b.dests[i].copy_(t) if b.dests[i] is not None else t
        lap("wait for the readers")
        if b.index + 1 < len(batches):               # the next batch's files start moving behind this batch's last copy: they arrive
            start(batches[b.index + 1])              # while this batch is on the device
        if timings is not None and timings.get("_synchronise"):          # developer timing only: separates the PCIe tail from the optimiser's time
            cur.synchronize()
            lap("h2d tail (timing runs only: synchronised)")
        heat_d = b.heat if b.heat is not None else (parts[0] if len(parts) == 1 else torch.cat(parts))
        b.pending = opt[0].fire(b.prep, heat_d, b.weights[0], b.weights[1], eps=eps, timings=timings)
        b.done = torch.cuda.Event()
        b.done.record(cur)
        lap("enqueue")

    def finish(b):
        """Batch b's results: waits for the device, merges / smooths / scores on it, fills the per-sequence lists."""
        # on a stream of its own, behind the batch's LAST kernel only: on the optimiser's stream the read-back would queue up
        # behind the next batch's whole device call, which is already enqueued there
        rs = _report_streams.get(device)
        if rs is None:
            rs = _report_streams[device] = torch.cuda.Stream(device=device, priority=STREAM_PRIORITY)
        rs.wait_event(b.done)
        for t in b.pending:
            if t is not None:
                t.record_stream(rs)
        with torch.cuda.stream(rs):
            _finish(b)

    def _finish(b):
        batch, counts = b.chunks, b.counts
        mid_local, opt_global, _ = opt[0].collect(b.pending, keep_device=True)
        b.pending = None
        lap("wait for the device + stats")
        mid_np = mid_local.cpu().numpy()
        lap("report: stage-one poses to the host")
        e = opt[0].engine
        if b.report is not None:
            # ... the error reports of all chunks are ONE library call and are read back with ONE synchronisation
            nb, r = len(batch), b.report
            A, X = r["mid_A"], np.ascontiguousarray(np.moveaxis(mid_np.astype(np.float64), -1, 0))          # X [3,W,T,J]
            mid_g = np.empty(mid_np.shape, dtype=np.float64)
            for d in range(3):
                mid_g[..., d] = A[d, 0] * X[0] + A[d, 1] * X[1] + A[d, 2] * X[2] + A[d, 3]
            mid_m = merge_chunks(mid_g, nb, overlap)
            fpc = r["est_m"].shape[1]
            lap("report: stage-one sequences (host float64)")
            opt_d = e.merge_windows(opt_global, nb, overlap=overlap, smooth=bool(final_smooth))          # [nb*fpc,15,3] f64, device
            mid_d = slots[b.index % N_BUFFERS][2].upload(mid_m.reshape(nb * fpc, 15, 3), torch.float64)
            reps = e.calculate_errors_chunks(r["est_d"], mid_d, opt_d, r["gt_d"], nb)
            lap("report: merge + error kernels enqueued")
            reps = reps.cpu().numpy()
            opt_m = opt_d.cpu().numpy().reshape(nb, fpc, 15, 3)
            lap("report: read-back")
            for k, c in enumerate(batch):
                res = OrderedDict(zip(e.ERROR_KEYS, reps[k, :17].tolist()))
                res["joints_error"] = reps[k, 17:].copy()
                gi = group_of[c["path"]]
                results[gi].append(res)
                raw_rows[gi].append(reps[k])
                est_all[gi].append(r["est_m"][k]); opt_all[gi].append(opt_m[k]); gt_all[gi].append(r["gt_m"][k])
                if verbose and res["bone_length_aligned_optimized_mpjpe"] > res["bone_length_aligned_mid_optimized_mpjpe"]:
                    print(res)
            lap("report: result dicts")
            return
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
            est_all[gi].append(np.asarray(est_seq)); opt_all[gi].append(np.asarray(opt_seq)); gt_all[gi].append(np.asarray(gt_seq))
            if verbose and res["bone_length_aligned_optimized_mpjpe"] > res["bone_length_aligned_mid_optimized_mpjpe"]:
                print(res)
        lap("sequences + reports")

    lap("plan")
    try:
        prev = None
        for b in batches:
            prepare(b)                    # while batch k-1 is on the device and batch k's files are arriving ...
            fire(b)                       # ... batch k is enqueued behind it as soon as its last file has been sent ...
            if prev is not None:
                finish(prev)              # ... and only then batch k-1's report is read back: the device never waits for the host
                prev.chunks = prev.reading = prev.parsing = prev.images = prev.report = prev.prep = None
            prev = b
        finish(prev)
    finally:
        _drain(submitted)
        if any(b.pending is not None for b in batches):
            torch.cuda.synchronize()      # (an exception left device work behind that reads this call's buffers)
    lap("readers drained")
    out = []
    for gi in range(n_groups):
        summary = OrderedDict()
        if len(raw_rows[gi]) == len(results[gi]):          # (every chunk of the sequence came as a row of the device report: one mean)
            mean = np.mean(np.stack(raw_rows[gi]), axis=0)
            for i, k in enumerate(results[gi][0]):
                summary[k] = mean[17:].copy() if k == "joints_error" else float(mean[i])
        else:
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
        # the three pose sequences as arrays [frames,15,3] (iterating them yields the [15,3] frames the reference's lists hold)
        out.append((summary, results[gi]) + tuple(np.concatenate(x) if x else np.empty((0, 15, 3)) for x in (est_all[gi], opt_all[gi], gt_all[gi])))
    lap("summaries")
    return out


def release_pools():
    """Give back what this module keeps between calls: the per-device frame buffers the readers fill, the pinned noise blocks
    and the reader threads (with their pinned staging buffers and device images).  Not to be called while another call is in
    flight."""
    _heat_pool.clear()
    _noise_pool.clear()
    _copy_streams.clear()
    _report_streams.clear()
    pools = list(_pools.values())
    _pools.clear()
    for p, _ in pools:
        p.shutdown(wait=True)
    if torch.cuda.is_available():
        torch.cuda.empty_cache()


def optimize_directory(data_dir, camera_model_path, *args, **kwargs):
    """One sequence = the reference's `optimize_whole_sequence.py`.  Returns (summary OrderedDict, per-chunk error
    dicts, estimated_pose, optimized_pose, gt_pose) -- the three pose sequences are the concatenations
    `optimize_whole_sequence.py:65-67` builds, as arrays [frames,15,3].  Arguments as `optimize_sequences`."""
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
