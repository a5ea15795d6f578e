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
