"""Developer probe (round 6): where the host time of the pickle path goes -- parse (gem_chunk_open) serial / threaded, staging alone,
noise alone, the report phases.  python tools/r06_ws_probe.py"""
import ctypes as C, os, pickle, sys, tempfile, time
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from globalegomocap_amd import synth, whole_sequence as ws, _capi

dev = torch.device("cuda:0")
torch.zeros(1, device=dev)
root = tempfile.mkdtemp(prefix="gem_probe_")
rng = np.random.default_rng(0)
heat = rng.random((100, 64, 64, 15), dtype=np.float32)
seq = {"estimated_local_skeleton": rng.random((100, 15, 3)), "gt_global_skeleton": rng.random((100, 15, 3)),
       "camera_pose_list": np.tile(np.eye(4), (100, 1, 1)), "heatmap_list": heat}
for i in range(20):
    d = os.path.join(root, "chunk_%d" % i); os.makedirs(d)
    with open(os.path.join(d, "test_data.pkl"), "wb") as f:
        pickle.dump(synth.reference_pickle_dict(seq), f)
paths = ws.list_chunks(root)
lib = _capi.load_library()
ck = (C.c_char_p * 4)(*[k.encode() for k in ws.KEYS])

def open_close(p):
    h = C.c_void_p()
    rc = lib.gem_chunk_open(os.fsencode(os.path.join(p, "test_data.pkl")), ck, 4, C.byref(h))
    assert rc == 0
    lib.gem_chunk_close(h)

def best(fn, n=7):
    b = 1e9
    for _ in range(n):
        t = time.perf_counter(); fn(); b = min(b, time.perf_counter() - t)
    return b * 1e3

print("gem_chunk_open+close serial, 20 chunks: %.2f ms" % best(lambda: [open_close(p) for p in paths]))
for th in (4, 8, 16):
    ex = ThreadPoolExecutor(th)
    list(ex.map(open_close, paths))
    print("gem_chunk_open+close, %d threads: %.2f ms" % (th, best(lambda: list(ex.map(open_close, paths)))))
    print("parse_chunk, %d threads: %.2f ms" % (th, best(lambda: [c.close() for c in ex.map(ws.parse_chunk, paths)])))
    ex.shutdown()
print("parse_chunk serial: %.2f ms" % best(lambda: [ws.parse_chunk(p).close() for p in paths]))

def stage_all(workers, depth=20):
    cs = list(ws.ChunkStream(paths, depth=depth, workers=workers, device=dev))
    torch.cuda.synchronize()
for w in (4, 8, 12, 16):
    ws.release_pools()
    stage_all(w)
    print("ChunkStream to device, %d workers: %.2f ms" % (w, best(lambda: stage_all(w))))
for sb in (1 << 20, 2 << 20, 8 << 20, 32 << 20):
    ws.SLICE_BYTES = sb
    print("  slice %d MB, 8 workers: %.2f ms" % (sb >> 20, best(lambda: stage_all(8))))
ws.SLICE_BYTES = 4 << 20
print("noise 480 x 2048 alone: %.2f ms" % best(lambda: ws._draw_noise([24] * 20, 2048)))
ex = ThreadPoolExecutor(1)
def both():
    f = ex.submit(ws._draw_noise, [24] * 20, 2048)
    stage_all(8)
    f.result()
print("noise + staging together: %.2f ms" % best(both))
