"""Developer probe (round 6): can the page cache feed the DMA engines directly?  mmap a chunk file, hipHostRegister the mapping,
hipMemcpyAsync from it -- against the product's route (pread into a pinned staging buffer, then the copy)."""
import ctypes as C, os, sys, tempfile, time
from concurrent.futures import ThreadPoolExecutor
import numpy as np
import torch
torch.zeros(1, device="cuda")
hip = C.CDLL([l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l][0])
libc = C.CDLL("libc.so.6", use_errno=True)
libc.mmap.restype = C.c_void_p
libc.mmap.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_long]
libc.munmap.argtypes = [C.c_void_p, C.c_size_t]
hip.hipHostRegister.argtypes = [C.c_void_p, C.c_size_t, C.c_uint]
hip.hipHostUnregister.argtypes = [C.c_void_p]
hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
root = tempfile.mkdtemp(prefix="probe_")
n, size = 20, 100 * 64 * 64 * 15 * 4
data = np.random.default_rng(0).random(size // 4, dtype=np.float32)
paths = []
for i in range(n):
    q = os.path.join(root, "f%d.bin" % i)
    with open(q, "wb") as f:
        f.write(data.tobytes())
    paths.append(q)
dst = torch.empty(n * size, dtype=torch.uint8, device="cuda")
stream = torch.cuda.Stream()
PROT_READ, MAP_SHARED, MAP_PRIVATE, MAP_POPULATE = 1, 1, 2, 0x8000
for name, mflags, rflags in (("MAP_SHARED, register default", MAP_SHARED, 0), ("MAP_SHARED|POPULATE, register default", MAP_SHARED | MAP_POPULATE, 0),
                             ("MAP_PRIVATE|POPULATE, register default", MAP_PRIVATE | MAP_POPULATE, 0), ("MAP_SHARED|POPULATE, register read-only (0x8)", MAP_SHARED | MAP_POPULATE, 8)):
    t_map = t_reg = t_unreg = 0.0
    ok = True
    def one(i):
        global ok
        fd = os.open(paths[i], os.O_RDONLY)
        t0 = time.perf_counter()
        p = libc.mmap(None, size, PROT_READ, mflags, fd, 0)
        t1 = time.perf_counter()
        rc = hip.hipHostRegister(p, size, rflags)
        t2 = time.perf_counter()
        if rc != 0:
            ok = False
            libc.munmap(p, size); os.close(fd)
            return rc, 0, 0, 0
        rc2 = hip.hipMemcpyAsync(dst.data_ptr() + i * size, p, size, 1, stream.cuda_stream)
        return (rc2, p, fd, (t1 - t0, t2 - t1))
    for th in (1, 8):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with ThreadPoolExecutor(th) as ex:
            res = list(ex.map(one, range(n)))
        t_issue = time.perf_counter() - t0
        torch.cuda.synchronize()
        t_all = time.perf_counter() - t0
        if not ok:
            print("%-48s: hipHostRegister failed, rc %s" % (name, res[0][0]))
            break
        tm = np.mean([r[3][0] for r in res]) * 1e3, np.mean([r[3][1] for r in res]) * 1e3
        t0 = time.perf_counter()
        for rc, p, fd, _ in res:
            hip.hipHostUnregister(p); libc.munmap(p, size); os.close(fd)
        t_un = time.perf_counter() - t0
        good = bool(torch.equal(dst[:size].view(torch.float32).cpu(), torch.from_numpy(data)))
        print("%-48s %d threads: issued %.1f ms, copies done %.1f ms = %.1f GB/s (mmap %.2f ms, register %.2f ms per file; unregister all %.1f ms) correct %s"
              % (name, th, t_issue * 1e3, t_all * 1e3, n * size / t_all / 1e9, tm[0], tm[1], t_un * 1e3, good))
