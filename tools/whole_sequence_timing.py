"""Developer measurement: the host-inclusive rate of the drop-in interface.

Writes a synthetic 2000-frame sequence as 20 chunk directories (the reference's `test_data.pkl` schema), then times
`whole_sequence.optimize_directory` end to end -- pickle reading, host->device copies of the heat-maps, the batched
optimisation, device merge / smoothing / error report -- against the device-resident rate `bench.py` reports.

    python tools/whole_sequence_timing.py [weights_cache.pt]
"""
import os
import pickle
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
from globalegomocap_amd import synth, vae as V, whole_sequence as ws
from globalegomocap_amd.camera import FisheyeCamera, DEFAULT_CALIBRATION
from globalegomocap_amd.optimizer import SequenceOptimizer

dev = torch.device("cuda")
shape = V.VAEShape()
cam = FisheyeCamera.from_json(DEFAULT_CALIBRATION)
cache = sys.argv[1] if len(sys.argv) > 1 else None
if cache == "structured":
    sd_l, sd_g = V.structured_state_dict(shape, 7), V.structured_state_dict(shape, 8, feature_offset=3.0)
elif cache and os.path.exists(cache):
    sd_l, _, sd_g, _ = torch.load(cache, weights_only=False)
else:
    sd_l, _ = bench.fit_weights(shape, 101, dev, 2000, False)
    sd_g, _ = bench.fit_weights(shape, 102, dev, 2000, True)

root = tempfile.mkdtemp(prefix="gem_seq_")
order = os.environ.get("GEM_WS_ORDER", "F")            # F: the reference's files (loadmat's Fortran order); C: C-ordered lists
hdt = np.float64 if os.environ.get("GEM_WS_F64") else np.float32
dirs = []
for si in range(3):
    seq = synth.make_sequence_device(2000, 1000 + si, dev, cam, cam_jitter=bench.CAM_JITTER)
    heat = seq["heat"].cpu().numpy()
    dirs.append(os.path.join(root, "seq_%d" % si))
    for c in range(20):
        sl = slice(c * 100, (c + 1) * 100)
        d = os.path.join(dirs[-1], "chunk_%d" % c)
        os.makedirs(d)
        obj = synth.reference_pickle_dict({"estimated_local_skeleton": seq["est_local_np"][sl], "gt_global_skeleton": seq["gt_global"][sl],
                                           "camera_pose_list": seq["cams_np"][sl], "heatmap_list": heat[sl]}, heat_dtype=hdt)
        if order == "C":
            obj["heatmap_list"] = [np.ascontiguousarray(h) for h in obj["heatmap_list"]]
        with open(os.path.join(d, "test_data.pkl"), "wb") as f:
            pickle.dump(obj, f)
size_mb = sum(os.path.getsize(os.path.join(dirs[0], d, "test_data.pkl")) for d in os.listdir(dirs[0])) / 1e6

if os.environ.get("GEM_WS_BENCHCTX"):          # (experiment: what bench.py has done to the process before its host-inclusive leg)
    from globalegomocap_amd.engine import WindowEngine, energy_weights
    from globalegomocap_amd.sequence import window_starts
    seqd = synth.make_sequence_device(2000, seed=1000, device=dev, camera=cam, cam_jitter=bench.CAM_JITTER)
    eng = WindowEngine(shape, cam, max_windows=240)
    eng.load_vae(0, sd_l); eng.load_vae(1, sd_g)
    st_ = np.concatenate([c * 100 + window_starts(100) for c in range(20)]).astype(np.int32)
    f0 = torch.as_tensor(st_, device=dev)
    mb = eng.mean_bone_length(seqd["est_local"][:100]).reshape(1, 15).expand(240, 15).contiguous()
    eps = torch.randn(240, 2, 2048)
    el, eg = eps[:, 0].contiguous().to(dev), eps[:, 1].contiguous().to(dev)
    wl, wg = energy_weights(1e-6, 1e-5, 1e-2, 0, 1e-2), energy_weights(1e-2, 1e-3, 1e-2, 0, 0)
    mode = os.environ["GEM_WS_BENCHCTX"]
    for i in range(60):
        if "p" in mode:
            eng.profile_enable(i < 2)
        eng.optimize_windows(seqd["est_local"], seqd["cams"], seqd["heat"], f0, mb, el, eg, wl, wg)
    torch.cuda.synchronize()
    if "p" in mode:
        eng.profile_enable(False); eng.profile_read(0); eng.profile_read(1); eng.profile_read(2)
    if "m" in mode:
        for m_ in ("bf16x3", "bf16", "f32"):
            eng.set_precision(m_)
            for i in range(10):
                eng.optimize_windows(seqd["est_local"], seqd["cams"], seqd["heat"], f0, mb, el, eg, wl, wg)
        torch.cuda.synchronize()
    if "c" in mode:
        eng.close(); del eng, seqd
        torch.cuda.empty_cache()
_burnt = [torch.cuda.Stream() for _ in range(int(os.environ.get("GEM_WS_BURN_STREAMS", 0)))]      # (experiment: other users of torch's stream pool before us)
opt = SequenceOptimizer(DEFAULT_CALIBRATION, sd_g, sd_l, max_windows=240)
print("pickles: %.0f MB in 20 chunks per sequence, heat-maps %s-ordered %s" % (size_mb, order, hdt.__name__))
for tag, fn, nw in (("one sequence (optimize_directory)", lambda tm: ws.optimize_directory(dirs[0], DEFAULT_CALIBRATION, optimizer=opt, verbose=False, timings=tm), 240),
                    ("one sequence, 2 batches of 10 chunks", lambda tm: ws.optimize_directory(dirs[0], DEFAULT_CALIBRATION, optimizer=opt, verbose=False, timings=tm, chunks_per_batch=10), 240),
                    ("three sequences pipelined (optimize_sequences per_sequence)", lambda tm: ws.optimize_sequences(dirs, DEFAULT_CALIBRATION, optimizer=opt, verbose=False, timings=tm, per_sequence=True), 720)):
    fn(None)      # warm-up
    torch.cuda.synchronize()
    t_read = time.perf_counter()
    chunks = list(ws.ChunkStream(ws.list_chunks(dirs[0]), device=dev, depth=20))
    torch.cuda.synchronize()
    t_read = time.perf_counter() - t_read
    del chunks
    runs = []
    for _ in range(9):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tm = {}
        res = fn(tm)
        torch.cuda.synchronize()
        runs.append(time.perf_counter() - t0)
        if runs[-1] == min(runs):
            best_tm = tm
    best = min(runs)
    summary = res[0] if isinstance(res, tuple) else res[0][0]
    print("   phases of the best run (ms):", {k: round(v * 1e3, 2) for k, v in best_tm.items() if not k.startswith("_")})
    print("   main thread's log of the best run (ms since the call began):", best_tm.get("_log"))
    print("%s: reading + upload of one sequence alone %.1f ms; end to end %.1f ms (best of 9; %s) = %.0f windows/s host-inclusive; "
          "optimized_global_mpjpe %.2f mm" % (tag, t_read * 1e3, best * 1e3, ", ".join("%.1f" % (r * 1e3) for r in runs), nw / best,
                                               summary["optimized_global_mpjpe"] * 1e3))
if os.environ.get("GEM_WS_PROFILE"):
    import cProfile, pstats
    pr = cProfile.Profile()
    pr.enable()
    ws.optimize_directory(dirs[0], DEFAULT_CALIBRATION, optimizer=opt, verbose=False)
    torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
