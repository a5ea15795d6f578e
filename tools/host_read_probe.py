"""Developer probe: how fast can this host move 20 x 24.6 MB page-cached files into pinned memory?  (loader design)"""
import mmap, os, sys, tempfile, time
from concurrent.futures import ThreadPoolExecutor
import numpy as np
import torch
root = tempfile.mkdtemp(prefix="probe_")
n, size = 20, 100 * 64 * 64 * 15 * 4
data = np.random.default_rng(0).random(size // 4, dtype=np.float32)
paths = []
for i in range(n):
    p = os.path.join(root, "f%d.bin" % i)
    with open(p, "wb") as f:
        f.write(data.tobytes())
    paths.append(p)
for p in paths:                      # warm the page cache
    open(p, "rb").read()
pinned = [torch.empty(size // 4, dtype=torch.float32).pin_memory() for _ in range(n)]
def rd(i):
    with open(paths[i], "rb", buffering=0) as f:
        f.readinto(memoryview(pinned[i].numpy()).cast("B"))
def mm(i):
    with open(paths[i], "rb") as f:
        m = mmap.mmap(f.fileno(), 0, prot=mmap.PROT_READ)
        np.copyto(pinned[i].numpy(), np.frombuffer(m, dtype=np.float32))
        m.close()
for name, fn in (("readinto", rd), ("mmap+copyto", mm)):
    for th in (1, 4, 8, 16):
        best = 1e9
        for _ in range(5):
            t0 = time.perf_counter()
            with ThreadPoolExecutor(th) as ex:
                list(ex.map(fn, range(n)))
            best = min(best, time.perf_counter() - t0)
        print("%-12s %2d threads: %6.1f ms = %5.1f GB/s" % (name, th, best * 1e3, n * size / best / 1e9))
dev = torch.device("cuda")
dst = torch.empty(n * size // 4, dtype=torch.float32, device=dev)
for _ in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        dst[i * (size // 4):(i + 1) * (size // 4)].copy_(pinned[i], non_blocking=True)
    torch.cuda.synchronize()
    print("H2D of the 20 pinned buffers: %.1f ms = %.1f GB/s" % ((time.perf_counter() - t0) * 1e3, n * size / (time.perf_counter() - t0) / 1e9))
print("cpus:", len(os.sched_getaffinity(0)))

# ---- the same copies issued from reader threads on their own streams (what whole_sequence.ChunkStream does)
import threading
def h2d_threaded(th, own_stream=True):
    streams = {}
    def work(i):
        tid = threading.get_ident()
        if tid not in streams:
            streams[tid] = torch.cuda.Stream() if own_stream else torch.cuda.current_stream()
        with torch.cuda.stream(streams[tid]):
            dst[i * (size // 4):(i + 1) * (size // 4)].copy_(pinned[i], non_blocking=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with ThreadPoolExecutor(th) as ex:
        list(ex.map(work, range(n)))
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    return t_host, time.perf_counter() - t0
for th in (1, 2, 4, 8):
    for own in (True, False):
        best = min(h2d_threaded(th, own) for _ in range(4))
        print("H2D from %d threads (%s): host %.1f ms, done %.1f ms = %.1f GB/s" % (th, "own streams" if own else "one stream", best[0] * 1e3, best[1] * 1e3, n * size / best[1] / 1e9))

# ---- read + copy pipeline: each thread reads a file into one of its two pinned buffers and copies it, like the loader
def pipeline(th, own_stream=True):
    local = threading.local()
    def work(i):
        if not hasattr(local, "s"):
            local.s = torch.cuda.Stream() if own_stream else torch.cuda.current_stream()
            local.buf = [torch.empty(size // 4, dtype=torch.float32).pin_memory() for _ in range(2)]
            local.ev = [None, None]; local.k = 0
        k = local.k; local.k ^= 1
        if local.ev[k] is not None:
            local.ev[k].synchronize()
        with open(paths[i], "rb", buffering=0) as f:
            f.readinto(memoryview(local.buf[k].numpy()).cast("B"))
        with torch.cuda.stream(local.s):
            dst[i * (size // 4):(i + 1) * (size // 4)].copy_(local.buf[k], non_blocking=True)
            ev = torch.cuda.Event(); ev.record(local.s)
        local.ev[k] = ev
    ex = ThreadPoolExecutor(th)
    for rep in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        list(ex.map(work, range(n)))
        t_host = time.perf_counter() - t0
        torch.cuda.synchronize()
        t = time.perf_counter() - t0
    ex.shutdown()
    return t_host, t
for th in (4, 8, 10, 20):
    for own in (True, False):
        a, b = pipeline(th, own)
        print("read+copy pipeline, %2d threads (%s): host %.1f ms, done %.1f ms = %.1f GB/s" % (th, "own streams" if own else "one stream", a * 1e3, b * 1e3, n * size / b / 1e9))
