"""Developer aid: where the time of reading 20 chunk pickles goes (plain un-pickling against the payload-skipping reader)."""
import os, pickle, sys, tempfile, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from globalegomocap_amd import synth, whole_sequence as ws

root = tempfile.mkdtemp(dir="/tmp")
rng = np.random.default_rng(0)
heat = rng.random((100, 64, 64, 15), dtype=np.float32)
for i in range(20):
    d = os.path.join(root, "chunk_%d" % i); os.makedirs(d)
    with open(os.path.join(d, "test_data.pkl"), "wb") as f:
        pickle.dump({"estimated_local_skeleton": list(rng.random((100, 15, 3))), "gt_global_skeleton": list(rng.random((100, 15, 3))),
                     "camera_pose_list": list(rng.random((100, 4, 4))), "heatmap_list": list(heat)}, f, protocol=4)
paths = ws.list_chunks(root)
dev = torch.device("cuda:0")
torch.zeros(1, device=dev)
for rep in range(3):
    t = time.perf_counter()
    for p in paths:
        r = ws._load_pickle_skipping(p)
        assert r is not None
        r[0].close()
    t_parse = time.perf_counter() - t
    t = time.perf_counter()
    for p in paths:
        with open(os.path.join(p, "test_data.pkl"), "rb") as f:
            pickle.load(f)
    t_plain = time.perf_counter() - t
    for workers in (1, 4, 8):
        t = time.perf_counter()
        got = list(ws.ChunkStream(paths, depth=8, workers=workers, device=dev))
        for c in got:
            c["heat_ready"].synchronize()
        t_stream = time.perf_counter() - t
        print("rep %d: parse-only %.1f ms, plain pickle.load %.1f ms, ChunkStream(%d workers) to device %.1f ms" % (rep, t_parse * 1e3, t_plain * 1e3, workers, t_stream * 1e3))

# reference points: the raw-array cache path, and the bare host-to-device copies
list(ws.ChunkStream(paths, depth=8, workers=8, device=dev, sidecar=True))
for rep in range(2):
    t = time.perf_counter()
    got = list(ws.ChunkStream(paths, depth=8, workers=8, device=dev, sidecar=True))
    for c in got:
        c["heat_ready"].synchronize()
    print("raw-array cache, 8 workers: %.1f ms" % ((time.perf_counter() - t) * 1e3))
pin = [torch.empty(100 * 64 * 64 * 15, dtype=torch.float32).pin_memory() for _ in range(4)]
dst = torch.empty(20, 100 * 64 * 64 * 15, dtype=torch.float32, device=dev)
streams = [torch.cuda.Stream() for _ in range(4)]
for rep in range(2):
    torch.cuda.synchronize()
    t = time.perf_counter()
    for i in range(20):
        with torch.cuda.stream(streams[i % 4]):
            dst[i].copy_(pin[i % 4], non_blocking=True)
    torch.cuda.synchronize()
    print("bare H2D of 20 x 24.6 MB from pinned memory: %.1f ms" % ((time.perf_counter() - t) * 1e3))

# single-thread decomposition of the skipping path
pinb = torch.empty(100, 64, 64, 15, dtype=torch.float32).pin_memory()
vw = pinb.numpy()
for rep in range(2):
    tp = tf = ts = 0.0
    for p in paths:
        t = time.perf_counter()
        f, shape, offs, small = ws._load_pickle_skipping(p)
        t1 = time.perf_counter()
        arrs = [np.asarray(small[k], dtype=np.float64) for k in ("estimated_local_skeleton", "gt_global_skeleton", "camera_pose_list")]
        t2 = time.perf_counter()
        nb = 64 * 64 * 15 * 4
        iov = []
        for i, o in enumerate(offs):
            gap = o - (offs[i - 1] + nb) if i else 0
            if gap:
                iov.append(memoryview(bytearray(gap)))
            iov.append(memoryview(vw[i]).cast("B"))
        t3 = time.perf_counter()
        os.preadv(f.fileno(), iov, offs[0])
        t4 = time.perf_counter()
        f.close()
        tp += t1 - t; ts += (t2 - t1) + (t3 - t2); tf += t4 - t3
    print("single thread, 20 chunks: parse+validate %.1f ms, small arrays + iovec %.1f ms, preadv %.1f ms" % (tp * 1e3, ts * 1e3, tf * 1e3))
    t = time.perf_counter()
    for p in paths:
        with open(os.path.join(p, ws.SIDE_CACHE), "rb", buffering=0) as f:
            f.seek(4096); f.readinto(memoryview(vw).cast("B"))
    print("single thread, 20 chunks: readinto of the cache files %.1f ms" % ((time.perf_counter() - t) * 1e3))

# timeline of the 8-worker run: per chunk (thread, parse start, parse end, staged + enqueued), ms since the start
import threading
log = []
_orig_parse, _orig_stage = ws._load_pickle_skipping, ws._stage_file_to_device
def parse(path):
    t = time.perf_counter(); r = _orig_parse(path); log.append((threading.get_ident() % 1000, "parse", t, time.perf_counter())); return r
def stage(*a, **k):
    t = time.perf_counter(); r = _orig_stage(*a, **k); log.append((threading.get_ident() % 1000, "stage", t, time.perf_counter())); return r
ws._load_pickle_skipping, ws._stage_file_to_device = parse, stage
for rep in range(2):
    del log[:]
    t0 = time.perf_counter()
    got = list(ws.ChunkStream(paths, depth=8, workers=8, device=dev))
    t_list = time.perf_counter()
    for c in got:
        c["heat_ready"].synchronize()
    t_end = time.perf_counter()
print("timeline: list() done at %.1f ms, device done at %.1f ms" % ((t_list - t0) * 1e3, (t_end - t0) * 1e3))
for th, what, a, b in sorted(log, key=lambda r: r[2]):
    print("  thread %3d %-5s %6.2f -> %6.2f ms (%.2f)" % (th, what, (a - t0) * 1e3, (b - t0) * 1e3, (b - a) * 1e3))
