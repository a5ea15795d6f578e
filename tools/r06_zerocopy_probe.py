"""Developer probe (round 6): the gather kernel reading the PAGE CACHE directly over PCIe -- mmap the chunk file, hipHostRegister the
mapping (mapped into the device's address space), hipHostGetDevicePointer, gem_heat_gather with that pointer as its image: no staging
copy, no image in HBM, one pass.  Against the product's route (pread -> pinned -> hipMemcpyAsync -> gather: ~10 ms per 20 files)."""
import ctypes as C, os, sys, tempfile, time
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from globalegomocap_amd import _capi
lib = _capi.load_library()
torch.zeros(1, device="cuda")
hip = C.CDLL([l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l][0])
libc = C.CDLL("libc.so.6", use_errno=True)
libc.mmap.restype = C.c_void_p
libc.mmap.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_long]
libc.munmap.argtypes = [C.c_void_p, C.c_size_t]
hip.hipHostRegister.argtypes = [C.c_void_p, C.c_size_t, C.c_uint]
hip.hipHostUnregister.argtypes = [C.c_void_p]
hip.hipHostGetDevicePointer.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.c_uint]
root = tempfile.mkdtemp(prefix="probe_")
n_files, n, H, W, J = 20, 100, 64, 64, 15
per = H * W * J * 4
stride = per + 61
size = 37 + n * stride + 64
rng = np.random.default_rng(0)
blob = rng.integers(0, 255, size, dtype=np.uint8)
ref = np.ascontiguousarray(np.stack([np.frombuffer(blob.tobytes(), dtype=np.float32, count=H * W * J, offset=37 + i * stride).reshape((H, W, J), order="F") for i in range(n)]))
paths = []
for i in range(n_files):
    q = os.path.join(root, "f%d.bin" % i)
    with open(q, "wb") as f:
        f.write(blob.tobytes())
    paths.append(q)
out = torch.empty(n_files * n, H, W, J, device="cuda")
offs = torch.as_tensor(37 + stride * np.arange(n), dtype=torch.int64, device="cuda")
streams = [torch.cuda.Stream() for _ in range(4)]
PROT_READ, MAP_SHARED, MAP_POPULATE = 1, 1, 0x8000
for flags in (2, 3):
    for th in (1, 4, 8):
        keep = []
        def one(i):
            fd = os.open(paths[i], os.O_RDONLY)
            p = libc.mmap(None, size, PROT_READ, MAP_SHARED | MAP_POPULATE, fd, 0)
            rc = hip.hipHostRegister(p, size, flags)
            if rc:
                return rc
            dp = C.c_void_p()
            rc = hip.hipHostGetDevicePointer(C.byref(dp), p, 0)
            if rc:
                return 1000 + rc
            st = streams[i % len(streams)]
            rc = lib.gem_heat_gather(dp, size - 64, C.c_void_p(offs.data_ptr()), n, H, W, J, 0, 1, C.c_void_p(out[i * n:].data_ptr()), C.c_void_p(st.cuda_stream))
            keep.append((p, fd))
            return rc
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with ThreadPoolExecutor(th) as ex:
            rcs = list(ex.map(one, range(n_files)))
        t_issue = time.perf_counter() - t0
        torch.cuda.synchronize()
        t_all = time.perf_counter() - t0
        good = all(r == 0 for r in rcs) and bool(np.array_equal(out[:n].cpu().numpy().view(np.uint32), ref.view(np.uint32))) and bool(np.array_equal(out[-n:].cpu().numpy().view(np.uint32), ref.view(np.uint32)))
        t0 = time.perf_counter()
        for p, fd in keep:
            hip.hipHostUnregister(p); libc.munmap(p, size); os.close(fd)
        t_un = time.perf_counter() - t0
        print("register flags %d, %d threads: issued %.1f ms, all heat-maps in HBM after %.1f ms = %.1f GB/s over PCIe (unregister %.1f ms) rc %s correct %s"
              % (flags, th, t_issue * 1e3, t_all * 1e3, n_files * n * per / t_all / 1e9, t_un * 1e3, sorted(set(rcs)), good))
