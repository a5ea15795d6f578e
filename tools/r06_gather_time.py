"""Developer measurement (round 6): the heat-map gather kernels alone -- 2000 frames (493 MB of float32 out) from an image of 20 chunk
files in HBM, Fortran / C order, float32 / float64 payloads; HIP events, best of 20."""
import ctypes as C, os, pickle, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from globalegomocap_amd import _capi
lib = _capi.load_library()
dev = torch.device("cuda")
rng = np.random.default_rng(0)
n, H, W, J = 2000, 64, 64, 15
for dt, item in ((0, 4), (1, 8)):
    for fortran in (1, 0):
        per = H * W * J * item
        stride = per + 61                                   # payloads at odd byte offsets, as in a pickle
        image = torch.randint(0, 255, (n * stride + 64,), dtype=torch.uint8, device=dev)
        offs = torch.as_tensor(37 + stride * np.arange(n), dtype=torch.int64, device=dev)
        out = torch.empty(n, H, W, J, device=dev)
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        best = 1e9
        for _ in range(20):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for c in range(20):                               # one launch per chunk of 100 frames, as whole_sequence does
                _capi.check(lib.gem_heat_gather(C.c_void_p(image.data_ptr()), image.numel() - 64, C.c_void_p(offs[100 * c:].data_ptr()), 100, H, W, J, dt, fortran,
                                                C.c_void_p(out[100 * c:].data_ptr()), st), lib)
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        one = 1e9
        for _ in range(20):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            _capi.check(lib.gem_heat_gather(C.c_void_p(image.data_ptr()), image.numel() - 64, C.c_void_p(offs.data_ptr()), n, H, W, J, dt, fortran,
                                            C.c_void_p(out.data_ptr()), st), lib)
            e1.record(); torch.cuda.synchronize()
            one = min(one, e0.elapsed_time(e1))
        moved = n * (per + H * W * J * 4)
        print("   the same 2000 frames in ONE launch: %.3f ms = %.2f TB/s (%.2f of 8 TB/s)" % (one, moved / one / 1e9, moved / one / 1e9 / 8.0))
        print("payload %s, %s order: 20 launches %.3f ms = %.2f TB/s of HBM traffic (%.0f MB read + %.0f MB written), %.2f of 8 TB/s"
              % (("float32", "float64")[dt], ("C", "Fortran")[fortran], best, moved / best / 1e9, n * per / 1e6, n * H * W * J * 4 / 1e6, moved / best / 1e9 / 8.0))
