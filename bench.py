#!/usr/bin/env python3
"""Throughput of the window optimiser on MI355X: optimised windows per second.

    python bench.py --gpus N --steps K --warmup W [--workload seq2k|w8192|<n_chunks>]

One "step" = one pass of the hot path over one batch of windows: local stage -> float64
relative-global transform -> global stage -> global pose (the loop body of the reference's `main()`,
optimizer.py:370-423) for every window of the rank's synthetic sequence, inputs resident in HBM.
Default workload = BASELINE.json configs[1]: one 2000-frame sequence = 20 chunks x 12 windows = 240
windows per GPU, fp32 (weak scaling: every rank gets its own sequence; with N > 1 the refined poses are
all-gathered over RCCL inside the step).

Rank 0 prints ONE JSON line (contract in the task statement) with `roofline` for the dominant kernel
(the decoder_input products, timed with HIP events on the launch stream inside the timed region) and
`cpu_baseline` (the PyTorch-CPU port of the reference timed on this box's host cores, rank 0, N=1).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MATRIX_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, spec
PEAK_BF16_MATRIX_TFLOPS = 2500.0    # dense bf16 MFMA, spec
PEAK_HBM_GBPS = 8000.0              # MI355X_MICROARCH.md: HBM3E ~8 TB/s
CHUNK = 100                         # frames per chunk directory (process_test_data.py:177-184)
LATENT_GAIN = float(os.environ.get("GEM_BENCH_LATENT_GAIN", "8"))   # latent gauge of the synthetic VAEs (see vae_torch.fit_vae)
PROFILE_STEPS = 2
CAM_JITTER = (0.3, 0.002)           # SLAM-like camera noise: 0.3 deg, 2 mm per frame (keeps the global stage busy)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=50)
    p.add_argument("--warmup", type=int, default=10)
    p.add_argument("--workload", default="seq2k", help="seq2k (20 chunks, 240 windows per GPU; the default, weak scaling) | w8192 | <number of "
                   "chunks> | configs3 (65536 windows sharded over the ranks, bf16, strong scaling) | configs4 (12499 overlapping windows of "
                   "one 100k-frame stream, block-cyclic shards, hipGraph replay, strong scaling)")
    p.add_argument("--lanes", type=int, default=None, help="gem_set_lanes: batches of at least this many windows run as two half-batches "
                   "half a round apart (0 = one lane always = the library default)")
    p.add_argument("--windows", type=int, default=0, help="configs3 / configs4: total number of windows (default 65536 / 12499)")
    p.add_argument("--block", type=int, default=None, help="windows per block of the block-cyclic shards (configs4: default 8; configs3: default "
                   "contiguous shards, a block size needs --activity, i.e. ONE stream); 0 = contiguous shards")
    p.add_argument("--activity", type=int, default=None, help="configs3 / configs4: seed of synth.activity_profile -- the stream alternates between "
                   "quiet stretches (windows that leave L-BFGS early) and busy ones, 500-5000 frames each; configs3 then shards ONE such stream "
                   "contiguously (instead of one i.i.d. stream per rank); --block 0 = contiguous shards for configs4 as well")
    p.add_argument("--emulate-ranks", type=int, default=0, help="configs3 / configs4 with --gpus 1: run the shards of N ranks one after "
                   "the other in this process (same shards, same calls, no collective): the reference the multi-rank result must equal bitwise")
    p.add_argument("--dump", default=None, help="configs3 / configs4: rank 0 saves the gathered poses (and the merged sequence) to this .npz")
    p.add_argument("--fit-steps", type=int, default=2000, help="Adam steps to fit the synthetic VAEs (untimed)")
    p.add_argument("--cpu-windows", type=int, default=12, help="windows of the CPU baseline sample (0 = skip)")
    p.add_argument("--no-extra", action="store_true", help="skip the bf16x3 / bf16 side measurements")
    p.add_argument("--in-flight", action="store_true", help="also measure 1-3 sequences in flight on separate streams, eager and graph replay "
                   "(`sequences_in_flight`; twelve entries, off by default to keep the line short)")
    p.add_argument("--no-partition", action="store_true", help="skip the full-size 8-way partitions of configs[3] / configs[4] (`partition8`)")
    p.add_argument("--full-record", default=None, help="also write the whole JSON line, indented, to this file")
    p.add_argument("--no-graphs", action="store_true", help="configs4: eager launches instead of hipGraph replay (diagnosis)")
    p.add_argument("--graphs", action="store_true", help="configs3: hipGraph replay as well (configs4 has it by default)")
    p.add_argument("--verbose", action="store_true", help="progress lines on stderr (sharded legs)")
    p.add_argument("--no-profile", action="store_true", help="no HIP-event timing of the dominant kernel")
    p.add_argument("--precision", default=None, choices=["f32", "bf16x3", "bf16"],
                   help="arithmetic of the wide decoder/encoder products (f32 = BASELINE configs[1])")
    p.add_argument("--vae", default="fit", choices=["fit", "fit-device", "structured"],
                   help="synthetic VAE weights: 'fit' = briefly fitted with Adam on synthetic motion (default, as in round 1; PyTorch "
                        "autograd); 'fit-device' = the same recipe through the HIP trainer (gem_trainer_*, SURVEY 8 f.4); "
                        "'structured' = vae.structured_state_dict, the deterministic well-conditioned VAEs of the full-size "
                        "reference golden (no fitting kernels: profiling runs)")
    p.add_argument("--weights-cache", default=None, help="torch file to load/store the fitted synthetic VAEs (keeps the "
                   "fitting kernels out of a rocprof trace)")
    a = p.parse_args()
    if a.precision is None:
        a.precision = "bf16" if a.workload in ("configs3", "configs4") else "f32"      # BASELINE.json: configs[3] and [4] are bf16
    return a


def fit_weights(shape, seed, device, steps, relative, on_device=False):
    """Untimed: synthetic well-conditioned VAE in the reference's checkpoint schema (on_device: fitted by the HIP trainer)."""
    from globalegomocap_amd import synth
    if on_device:
        from globalegomocap_amd.vae_train import fit_vae_device as fit_vae
    else:
        from globalegomocap_amd.vae_torch import fit_vae
    win = synth.make_training_windows(4096, shape.seq_len, seed)
    if relative:        # relative-global poses drift with the (true) camera: 4 mm / frame along x
        win = win.reshape(-1, shape.seq_len, 15, 3).copy()
        win[..., 0] += (0.004 * np.arange(shape.seq_len))[None, :, None]
        win = win.reshape(-1, shape.seq_len, 45)
    sd, err = fit_vae(shape, win, steps=steps, batch=128, lr=2e-3, kl_weight=0.01, seed=seed, device=device,
                      latent_gain=LATENT_GAIN)
    return sd, err


def cpu_baseline(sd_local, sd_global, cam, seqd, starts, mean_bone, eps_l, eps_g, w_local, w_global, n_windows):
    """The reference's algorithm on the host cores (oracle/torch_port.py), first n_windows windows."""
    import torch
    from oracle import torch_port as TP
    from oracle import np_oracle as O
    # the box exposes all host threads but a 1-GPU job owns a 16-core share: do not oversubscribe
    nthreads = max(1, min(len(os.sched_getaffinity(0)), 16))
    torch.set_num_threads(nthreads)
    nets = [TP.vae_from_state_dict(sd_local), TP.vae_from_state_dict(sd_global)]
    est, cams = seqd["est_local_np"], seqd["cams_np"]
    heat = seqd["heat"][: int(starts[n_windows - 1]) + 10].cpu().numpy()
    opts = []
    for net, w in zip(nets, (w_local, w_global)):
        o = TP.WindowOptimizerPort(net, cam.poly_w2c, cam.cx, cam.cy, est[:CHUNK])
        o.mean_bone = torch.as_tensor(mean_bone)
        o.set_weights(*w)
        opts.append(o)
    out = []
    t0 = time.perf_counter()
    for i in range(n_windows):
        s = int(starts[i])
        loc, cs, hs = est[s:s + 10], cams[s:s + 10], heat[s:s + 10]
        a, _ = opts[0].optimize(loc, hs, eps_l[i])
        rel = O.relative_global(a, cs)
        b, _ = opts[1].optimize(rel.astype(np.float32), hs, eps_g[i])
        out.append(O.to_global(b, cs))
    dt = time.perf_counter() - t0
    return np.asarray(out), dt, nthreads


def committed_traffic(kernel_names, windows, precision, with_mfma=False):
    """HBM bytes per launch of the named kernels from the committed rocprofv3 --pmc summaries (profiles/traffic_r*.json: FETCH_SIZE x 2
    on gfx950 + WRITE_SIZE, separate passes -- MI355X_MICROARCH.md section HBM), dispatch-weighted over the names.  Only a file
    collected on THIS workload (same window count and precision: its `workload` header, or -- files of rounds 1-3 -- its name)
    qualifies, newest round first; a kernel the matching files do not know gives (None, None): another workload's bytes are never
    reported.  NOT measured in this run: the source file is named next to the number."""
    import glob
    import re
    names = [n.strip() for n in kernel_names.split(";") if n.strip()]

    def workload_of(path, d):
        w = d.get("workload")
        if isinstance(w, dict):
            return int(w.get("windows", -1)), str(w.get("precision", ""))
        m = re.search(r"traffic_r\d+_(f32|bf16)_(\d+)_windows", os.path.basename(path))
        if m:
            return int(m.group(2)), m.group(1)
        if re.fullmatch(r"traffic_r\d+\.json", os.path.basename(path)):
            return 240, "f32"                              # the default bench command (BASELINE configs[1])
        return -1, ""
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "traffic_r*.json")), reverse=True):
        try:
            d = json.load(open(path))
        except (OSError, ValueError):
            continue
        wn, wp = workload_of(path, d)
        # (8196 = the 683-chunk stand-in of rounds 2-3 for the 8192-window shard: same kernels, same tiles)
        if wp != precision or not (wn == windows or (windows == 8192 and wn == 8196)):
            continue
        k = d.get("kernels", {})
        hit = [k[n] for n in names if n in k]
        if len(hit) == len(names) and hit:
            # one timed "launch" of bench.py = one call of the library's launcher.  A call may enqueue kernels of TWO templates (the
            # one-round and the 128 x 128 front GEMM: one of them returns at once, decided on the device): per call = everything the
            # named kernels moved / the calls = the dispatches of the most-dispatched template
            fam = {}
            for n, h_ in zip([n for n in names if n in k], hit):
                fam[n.split("<")[0]] = fam.get(n.split("<")[0], 0) + h_["dispatches"]
            disp = max(fam.values())
            tot = sum((h["read_bytes_corrected"] + h["write_bytes"]) * h["dispatches"] for h in hit)
            src = "profiles/%s (committed --pmc passes, %d windows %s)" % (os.path.basename(path), wn, wp)
            if with_mfma:
                mf = None
                if all("mfma_busy_cycles" in h for h in hit):
                    famm = {}
                    for n, h_ in zip([n for n in names if n in k], hit):
                        famm[n.split("<")[0]] = famm.get(n.split("<")[0], 0) + h_.get("mfma_dispatches", h_["dispatches"])
                    dm = max(famm.values())
                    mf = {"busy": sum(h["mfma_busy_cycles"] * h.get("mfma_dispatches", h["dispatches"]) for h in hit) / dm,
                          "active": sum(h["gui_active_cycles"] * h.get("mfma_dispatches", h["dispatches"]) for h in hit) / dm}
                return int(tot / disp), src, mf
            return int(tot / disp), src
    return (None, None, None) if with_mfma else (None, None)


SIMDS = 1024                                        # 256 CUs x 4
FLOP_PER_SIMD_CYCLE = {"f32": PEAK_F32_MATRIX_TFLOPS * 1e12 / (SIMDS * 2.4e9), "bf16": PEAK_BF16_MATRIX_TFLOPS * 1e12 / (SIMDS * 2.4e9)}


def mfma_busy_record(mf, avg_us, flop_per_launch, operand):
    """SQ_VALU_MFMA_BUSY_CYCLES of the committed PMC pass (summed over the 1024 SIMDs, per launch) beside this run's timing:
    `counter_over_flop` = counter / (FLOP per launch / FLOP per SIMD-cycle of the MFMA shape) -- 1.0 means the counter and the FLOP
    count agree (above 1: padded tiles); `frac_at_peak_clock` = busy cycles per SIMD / (this run's average duration x 2.4 GHz),
    the counter's version of FLOP / time / peak; `frac_of_active_cycles` = busy / (1024 x GRBM_GUI_ACTIVE / 8) of the same PMC
    pass: utilisation at the clock the chip really held (the profiled dispatch adds ~15-20 k cycles, so it reads low for launches
    under ~50 us)."""
    if not mf:
        return None
    exp = flop_per_launch / FLOP_PER_SIMD_CYCLE[operand]
    return {"busy_simd_cycles_per_launch": round(mf["busy"]), "counter_over_flop": round(mf["busy"] / exp, 3) if exp > 0 else None,
            "frac_at_peak_clock": round(mf["busy"] / SIMDS / (avg_us * 2400.0), 4),
            "frac_of_active_cycles": round(mf["busy"] / (SIMDS * mf["active"]), 4)}


# SURVEY.md 8d, per stage with E evaluations: encoder + E x (decoder forward + backward-data) + the final decode, dense counts of the
# reference's layers (decoder 15 978 880 multiply-adds, encoder 26 341 760).  EXECUTED: decoder_input and the first decoder conv run as
# one composed [2048 -> 10 x 256] layer (2 x 2048 x 2560 flop) in front of the narrow convs (10 x 3 x 52 032 multiply-adds).
F_ENC, F_DEC = 2 * 26341760, 2 * 15978880
F_DEC_EXEC = 2 * 2048 * 2560 + 2 * 30 * (256 * 128 + 128 * 64 + 64 * 64 + 64 * 64 + 64 * 45)


def path_roofline(windows_per_s, e_local, e_global, precision):
    """The path-level roofline of SURVEY.md 8d.  `frac_executed` is the roofline fraction: windows/s x the FLOPs this implementation
    EXECUTES per window (measured mean evaluations per stage) / matrix peak.  `speedup_vs_dense_count` prices the same rate with
    SURVEY 8d's dense count of the reference's layers instead: the composed front layer executes 44 % of those FLOPs, so this
    figure may exceed 1 -- it says how much faster the path runs than a dense evaluation at peak could, not how busy the
    matrix units are."""
    alg = sum(F_ENC + 2 * F_DEC * e + F_DEC for e in (e_local, e_global))
    exe = sum(F_ENC + 2 * F_DEC_EXEC * e + F_DEC_EXEC for e in (e_local, e_global))
    peak = PEAK_BF16_MATRIX_TFLOPS if precision == "bf16" else PEAK_F32_MATRIX_TFLOPS
    return {"evals_per_stage": [round(float(e_local), 2), round(float(e_global), 2)],
            "flop_per_window_algorithmic": int(alg), "flop_per_window_executed": int(exe),
            "executed_tflops": round(windows_per_s * exe / 1e12, 2), "frac_executed": round(windows_per_s * exe / 1e12 / peak, 4),
            "dense_count_tflops": round(windows_per_s * alg / 1e12, 2), "speedup_vs_dense_count": round(windows_per_s * alg / 1e12 / peak, 4),
            "peak": peak, "unit": "TFLOP/s"}


def kernel_bound(names):
    """What bounds a timed kernel family, by the kernels that actually ran: the few-rows / tiled / LDS-DMA
    products are priced against the matrix peak; the fused tails are chains of short dependent phases (their matrix work is a
    fraction of their time): `latency`, the TFLOP/s figure is kept for comparison only."""
    if "decoder_tail" in names:
        return "latency"
    return "mfma"


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start N fresh ranks (one per GPU) under torch.distributed.run and
    relay their exit code.  Runs before this process touches torch.cuda or HIP: the parent only waits."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC only on this pool (RCCL needs it)
    return subprocess.run(cmd, env=env).returncode


def run_sharded(a, world, rank, device, rehearsal, sd_local, sd_global, vae_note, emit=True):
    """BASELINE configs[3] / configs[4]: ONE job of n windows sharded over the ranks (strong scaling).

    configs3: 65 536 independent windows, contiguous shards (`dist.shard_range`), every rank synthesises its own frames on its
      device from seed + rank (nothing is scattered from a root), bf16 decoder; one `all_gather_windows` of the refined poses.
    configs4: the 12 499 overlapping windows (stride 8) of ONE 100 000-frame stream, block-cyclic shards
      (`dist.shard_indices(block=k)`: blocks of k windows = 8k + 2 contiguous frames dealt round-robin, so that stretches whose
      windows exit early or late are spread over the ranks); a rank holds only the frames of its blocks (stored once, + the 2-frame
      halo between blocks that went to different ranks); every call after the second is one hipGraph replay;
      `all_gather_indexed` restores window order, then overlap-merge + final smoothing (optimizer.py:425-450) of the gathered
      sequence on the device.
    Real ranks (world > 1): the timed step = optimise the shard + the collective (+ merge for configs4); value = n windows / step time.
    --emulate-ranks N on one card: the N shards are built, warmed, timed and FREED one after the other (one shard of 8192 windows
      holds 16 GB of heat-maps: only one is resident at a time), every shard exactly as its rank would run it; then the placement by
      index (what the all-gather does) and the merge are timed.  `value` = n windows / (sum of the shard times + placement + merge):
      what this ONE card did.  `partition` reports per emulated rank windows, frames held, evaluations and time, the imbalance
      max/mean of both, and `projected_8gpu_windows_per_s` = n / (max_r t_r + placement + merge) -- a PROJECTION of the N-GPU job
      from one card's shard times (the RCCL all-gather of 1.8 KB per window is not in it), not a measurement.
    emit=False: rank 0 returns the record instead of printing it (side records of the default line), the process group stays up."""
    import torch
    import torch.distributed as dist
    from globalegomocap_amd import synth, vae as vae_schema
    from globalegomocap_amd.camera import FisheyeCamera, DEFAULT_CALIBRATION
    from globalegomocap_amd.dist import shard_indices, frame_runs, all_gather_windows, all_gather_indexed
    from globalegomocap_amd.engine import WindowEngine, energy_weights, stats_to_numpy, LOCAL_STAGE, GLOBAL_STAGE
    from globalegomocap_amd.errors import mpjpe
    stream = a.workload == "configs4" or a.activity is not None          # (with an activity profile configs3 is one long stream too)
    n_total = a.windows or (12499 if a.workload == "configs4" else 65536)
    # configs4: blocks of 8 windows (66 + 2 halo frames) dealt round-robin: on a recording with quiet and busy stretches the
    # evaluations per rank are within 1.2 % of one another (blocks of 32: 5.8 %, contiguous shards: 17 %)
    block = (8 if a.block is None else (a.block or None)) if a.workload == "configs4" else ((a.block or None) if stream else None)
    emulated = bool(a.emulate_ranks and world == 1)
    vworld = a.emulate_ranks if emulated else world
    my_ranks = list(range(vworld)) if emulated else [rank]
    shape = vae_schema.VAEShape()
    cam = FisheyeCamera.from_json(DEFAULT_CALIBRATION)
    T, stride = shape.seq_len, 8
    idx_of = {r: shard_indices(n_total, r, vworld, block) for r in range(vworld)}
    cap = max(len(v) for v in idx_of.values())
    eng = WindowEngine(shape, cam, max_windows=max(cap, 1))
    eng.load_vae(LOCAL_STAGE, sd_local)
    eng.load_vae(GLOBAL_STAGE, sd_global)
    eng.set_precision(a.precision)
    if a.lanes is not None:
        eng.set_lanes(a.lanes)
    eng.enable_graphs((a.workload == "configs4" or getattr(a, "graphs", False)) and not getattr(a, "no_graphs", False))

    def say(*msg):
        if getattr(a, "verbose", False):
            print("[bench rank %d]" % rank, *msg, file=sys.stderr, flush=True)
    wl = (0.01 / 10000, 0.001 / 100, 0.01, 0.0, 0.01)
    wg = (0.01, 0.001, 0.01, 0.0, 0.0)
    w_local, w_global = energy_weights(*wl), energy_weights(*wg)
    pose_shape = (T, 15, 3)
    host_stream = None
    if stream:
        n_frames = stride * (n_total - 1) + T
        starts = (stride * np.arange(n_total)).astype(np.int64)
        eps_all = torch.randn(n_total, 2, shape.latent_dim, generator=torch.Generator().manual_seed(654))
        act = synth.activity_profile(n_frames, a.activity) if a.activity is not None else None
        host_stream = synth.make_sequence(n_frames, 6000, cam, with_heatmaps=False, cam_jitter=CAM_JITTER, activity=act)    # same stream on every rank

    def build_shard(r):
        idx = idx_of[r]
        nb = len(idx)
        if stream:
            runs, local0 = frame_runs(starts, idx, T)
            d = synth.make_stream_device(n_frames, 6000, device, runs=runs, camera=cam, cam_jitter=CAM_JITTER, host=host_stream)
            mbv = eng.mean_bone_length(d["est_all_np"].astype(np.float32))       # one sequence: one mean bone length (all ranks: same stream)
            eps = eps_all[torch.as_tensor(idx, dtype=torch.long)] if nb else eps_all[:0]
        else:
            d = synth.make_stream_device(stride * max(nb - 1, 0) + T, 5000 + r, device, camera=cam, cam_jitter=CAM_JITTER)
            local0 = (stride * np.arange(nb)).astype(np.int32)
            mbv = eng.mean_bone_length(d["est_local"])
            eps = torch.randn(nb, 2, shape.latent_dim, generator=torch.Generator().manual_seed(7000 + r))
        return dict(d=d, idx=idx, f0=torch.as_tensor(local0, dtype=torch.int32, device=device),
                    mb=mbv.reshape(1, 15).expand(nb, 15).contiguous(), el=eps[:, 0].contiguous().to(device),
                    eg=eps[:, 1].contiguous().to(device), frames_held=int(d["heat"].shape[0]))

    def run_shard(sh):
        if len(sh["idx"]) == 0:
            return torch.zeros((0,) + pose_shape, dtype=torch.float64, device=device), None
        _, glob, stats = eng.optimize_windows(sh["d"]["est_local"], sh["d"]["cams"], sh["d"]["heat"], sh["f0"], sh["mb"], sh["el"], sh["eg"],
                                              w_local, w_global, want_stats=True)
        return glob, stats

    def place(outs_by_rank, dtype=torch.float64, trailing=pose_shape):
        """Emulated ranks: the placement by window index that the all-gather performs, without a collective."""
        full = torch.empty((n_total,) + trailing, dtype=dtype, device=device)
        for r in my_ranks:
            if len(idx_of[r]):
                full[torch.as_tensor(idx_of[r], dtype=torch.long, device=device)] = outs_by_rank[r]
        return full

    def shard_accuracy(sh, glob):
        """configs3 (every shard is its own stride-8 stream): MPJPE of the merged + smoothed shard and of its input, in mm."""
        d = sh["d"]
        m = eng.merge_windows(glob, 1, overlap=T - stride, smooth=True).cpu().numpy()
        gt = d["gt_global"][:m.shape[0]]
        homo = np.concatenate([d["est_all_np"][:m.shape[0]], np.ones((m.shape[0], 15, 1))], -1)
        est_g = np.einsum("nij,nkj->nki", d["cams"][:m.shape[0]].cpu().numpy(), homo)[..., :3]
        return mpjpe(m, gt) * 1e3, mpjpe(est_g, gt) * 1e3

    n_warm = max(a.warmup, 3 if eng._graphs else 1)          # (graphs: eager, capture, first replay)
    outs, stats_of, per_rank, acc = {}, {}, [], []
    merged = None
    if emulated:
        # ---- one card plays the N ranks one after the other; one shard resident at a time
        t_shards = []
        for r in my_ranks:
            sh = build_shard(r)
            say("emulated rank", r, "windows", len(sh["idx"]), "frames", sh["frames_held"], "ptrs", [hex(sh["d"][k].data_ptr()) for k in ("est_local", "cams", "heat")],
                [hex(sh[k].data_ptr()) for k in ("f0", "mb", "el", "eg")])
            for i in range(n_warm):
                run_shard(sh)
                say("  warm-up", i, "enqueued", eng.graph_stats() if stream else "")
            torch.cuda.synchronize()
            t_steps = []
            for _ in range(a.steps):          # (every step timed by itself: the imbalance statistic takes each rank's BEST step, so that
                t0 = time.perf_counter()      # a one-off stall -- first-touch allocations, a clock ramp -- does not read as imbalance)
                glob, stats = run_shard(sh)
                torch.cuda.synchronize()
                t_steps.append(time.perf_counter() - t0)
                say("  step done %.3f ms" % (t_steps[-1] * 1e3))
            t_r = sum(t_steps) / a.steps
            outs[r] = glob.clone()
            stats_of[r] = stats_to_numpy(stats) if stats is not None else None
            if not stream and len(sh["idx"]):
                acc.append(shard_accuracy(sh, glob))
            t_shards.append(t_r)
            per_rank.append({"rank": r, "windows": int(len(sh["idx"])), "frames_held": sh["frames_held"],
                             "evaluations": int(stats_of[r]["func_evals"].sum()) if stats_of[r] is not None else 0,
                             "ms_per_step": round(t_r * 1e3, 3), "ms_best_step": round(min(t_steps) * 1e3, 3)})
            del sh, glob, stats
            if eng._graphs:
                # the captured calls hold the addresses of this shard's tensors: drop them before the memory goes back to the driver
                # (the next shard's tensors may land on the same addresses; a replay of a graph captured before the free / re-allocation
                # ended in a GPU memory fault on ROCm 7.2 -- DESIGN.md section 7)
                eng.drop_graphs()
            torch.cuda.empty_cache()
        for _ in range(2):
            full = place(outs)
            merged = eng.merge_windows(full, 1, overlap=T - stride, smooth=True) if stream else None
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            full = place(outs)
            merged = eng.merge_windows(full, 1, overlap=T - stride, smooth=True) if stream else None
        torch.cuda.synchronize()
        t_post = (time.perf_counter() - t0) / a.steps
        elapsed = (sum(t_shards) + t_post) * a.steps
        order = place({r: torch.as_tensor(idx_of[r], dtype=torch.int64, device=device) for r in my_ranks}, torch.int64, ())
    else:
        sh = build_shard(rank)

        def step():
            glob, stats = run_shard(sh)
            if world > 1:          # the one collective of the path
                full = all_gather_indexed(glob, n_total, block) if stream else all_gather_windows(glob, n_total)
            else:
                full = glob
            return full, (eng.merge_windows(full, 1, overlap=T - stride, smooth=True) if stream else None), glob, stats

        for _ in range(n_warm):
            step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            full, merged, glob, stats = step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        t_mine = elapsed / a.steps
        if world > 1:
            t = torch.tensor([elapsed], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        outs[rank] = glob
        stats_of[rank] = stats_to_numpy(stats) if stats is not None else None
        if not stream and len(sh["idx"]):
            acc.append(shard_accuracy(sh, glob))
        mine = {"rank": rank, "windows": int(len(sh["idx"])), "frames_held": sh["frames_held"],
                "evaluations": int(stats_of[rank]["func_evals"].sum()) if stats_of[rank] is not None else 0,
                "ms_per_step": round(t_mine * 1e3, 3)}
        my_idx = torch.as_tensor(idx_of[rank], dtype=torch.int64, device=device)
        if world > 1:
            per_rank = [None] * world
            dist.all_gather_object(per_rank, mine)
            acc_all = [None] * world
            dist.all_gather_object(acc_all, acc)
            acc = [x for part in acc_all for x in part]
            order = all_gather_indexed(my_idx, n_total, block) if stream else all_gather_windows(my_idx, n_total)
        else:
            per_rank = [mine]
            order = my_idx
        t_post = None
    fin = all(bool(s["finished"].all()) for s in stats_of.values() if s is not None)
    if world > 1:
        fl = [None] * world
        dist.all_gather_object(fl, fin)
        fin = all(fl)
    order_ok = bool(torch.equal(order.cpu(), torch.arange(n_total, dtype=torch.int64)))
    line = None
    if rank == 0:
        mp = mp_in = None
        if stream:
            gt_all = np.asarray(host_stream["gt_global_skeleton"])
            nm = merged.shape[0]
            mp = round(mpjpe(merged.cpu().numpy(), gt_all[:nm]) * 1e3, 3)
            homo = np.concatenate([np.asarray(host_stream["estimated_local_skeleton"])[:nm], np.ones((nm, 15, 1))], -1)
            est_g = np.einsum("nij,nkj->nki", np.asarray(host_stream["camera_pose_list"])[:nm], homo)[..., :3]
            mp_in = round(mpjpe(est_g, gt_all[:nm]) * 1e3, 3)
        elif acc:
            mp, mp_in = round(float(np.mean([x[0] for x in acc])), 3), round(float(np.mean([x[1] for x in acc])), 3)
        if a.dump:
            st0 = stats_of[my_ranks[0]]
            np.savez(a.dump, glob=full.cpu().numpy(), merged=merged.cpu().numpy() if merged is not None else np.zeros(0),
                     **({"stats_" + k: np.asarray(st0[k]) for k in st0.dtype.names} if st0 is not None else {}))
        gs = eng.graph_stats() if stream else None
        ev = [float(p["evaluations"]) for p in per_rank]
        tm = [float(p.get("ms_best_step", p["ms_per_step"])) for p in per_rank]
        t_max = max(tm) * 1e-3
        partition = {"per_rank": per_rank,
                     "evaluations_per_rank": {"max_over_mean": round(max(ev) / max(1e-9, sum(ev) / len(ev)), 4)},
                     "time_per_rank": {"max_over_mean": round(max(tm) / max(1e-9, sum(tm) / len(tm)), 4),
                                       "of": "each rank's best step" if emulated else "mean step"},
                     "placement_and_merge_ms": round(t_post * 1e3, 3) if t_post is not None else None,
                     "gathered_order_is_arange": order_ok}
        if emulated:
            partition["projected_%dgpu_windows_per_s" % vworld] = round(n_total / (t_max + t_post), 1)
            partition["projection_note"] = ("PROJECTION, not a measurement: n windows / (slowest emulated rank's best step + placement + merge), "
                                            "each shard run alone on ONE card; the RCCL all-gather (1.8 KB per window) is not included")
        line = {
            "metric": "optimised windows/sec (10-frame, 15-joint)", "value": round(n_total * a.steps / elapsed, 2), "unit": "windows/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(elapsed / a.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": {"f32": "f32", "bf16x3": "f32 via 3x bf16 split MFMA", "bf16": "bf16 decoder products and activations / f32 accumulate, energies, L-BFGS"}[a.precision],
            "data": "synthetic",
            "config": {"workload": ("BASELINE configs[4]: %d overlapping windows (stride 8) of ONE %d-frame stream, block-cyclic shards of %d-window "
                                    "blocks over %d ranks, every rank holds only its blocks' frames (+ halo), hipGraph replay of the whole call, "
                                    "all-gather by index, overlap-merge + final smoothing of the gathered sequence"
                                    % (n_total, stride * (n_total - 1) + T, block or 0, vworld)) if stream else
                                   ("BASELINE configs[3]: %d independent windows in contiguous shards over %d ranks (%d per rank), every rank "
                                    "synthesises its own frames from seed + rank, one all-gather of the refined poses" % (n_total, vworld, cap)),
                       "activity_profile": (None if a.activity is None else "synth.activity_profile(seed %d): quiet and busy stretches of 500-5000 frames; "
                                            "%s shards" % (a.activity, "contiguous" if not block else "block-cyclic (%d windows)" % block)),
                       "windows_total": n_total, "windows_per_rank_max": cap, "ranks": world, "emulated_ranks": vworld if emulated else None,
                       "backend": dist.get_backend() if world > 1 else None, "rehearsal_on_one_card": bool(rehearsal),
                       "vae": vae_note, "precision": a.precision,
                       "collective": None if world == 1 else ("all_gather_indexed" if stream else "all_gather_windows")},
            "all_finished": fin, "mpjpe_input_mm": mp_in, "mpjpe_optimised_mm": mp, "graph": gs,
            "evaluations_per_rank": {"sum": [round(v) for v in ev], "max_over_mean": partition["evaluations_per_rank"]["max_over_mean"]},
            "partition": partition,
        }
        if emit:
            print(json.dumps(line), flush=True)
    eng.close()
    outs.clear()
    torch.cuda.empty_cache()
    if world > 1:
        dist.barrier()
        if emit:
            dist.destroy_process_group()
    return line


def main():
    a = parse()
    if a.gpus < 1:
        sys.exit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        sys.exit(spawn_ranks(a.gpus))
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        sys.exit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (a.gpus, world))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # GEM_BENCH_REHEARSAL=1: several ranks on ONE card with gloo, to rehearse the multi-rank code path on a 1-GPU box
    rehearsal = os.environ.get("GEM_BENCH_REHEARSAL") == "1"
    if not rehearsal and torch.cuda.device_count() < world:
        sys.exit("bench.py: %d ranks but only %d GPUs visible (GEM_BENCH_REHEARSAL=1 shares one card over gloo)"
                 % (world, torch.cuda.device_count()))
    dev_index = local_rank % torch.cuda.device_count() if rehearsal else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1:
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)       # RCCL over xGMI

    import __graft_entry__ as ge
    if rank == 0:
        ge.build()
    if world > 1:
        dist.barrier()
    from globalegomocap_amd import synth, vae as vae_schema
    from globalegomocap_amd.camera import FisheyeCamera, DEFAULT_CALIBRATION
    from globalegomocap_amd.engine import WindowEngine, energy_weights, stats_to_numpy, LOCAL_STAGE, GLOBAL_STAGE
    from globalegomocap_amd.sequence import window_starts, merge_batches, final_smooth
    from globalegomocap_amd.errors import mpjpe

    n_chunks = {"seq2k": 20, "w8192": 683, "w8192x": 683, "configs3": 0, "configs4": 0}.get(a.workload)
    if n_chunks is None:
        n_chunks = int(a.workload)
    shape = vae_schema.VAEShape()
    cam = FisheyeCamera.from_json(DEFAULT_CALIBRATION)

    # ---- untimed setup: weights (same on every rank), the rank's own sequence, window table
    if a.vae == "structured":
        sd_local = vae_schema.structured_state_dict(shape, 7, feature_offset=0.0)
        sd_global = vae_schema.structured_state_dict(shape, 8, feature_offset=3.0)
        err_l = err_g = float("nan")
    elif a.weights_cache and os.path.exists(a.weights_cache):
        sd_local, err_l, sd_global, err_g = torch.load(a.weights_cache, weights_only=False)
    else:
        sd_local, err_l = fit_weights(shape, 101, device, a.fit_steps, relative=False, on_device=a.vae == "fit-device")
        sd_global, err_g = fit_weights(shape, 102, device, a.fit_steps, relative=True, on_device=a.vae == "fit-device")
        if a.weights_cache and rank == 0:
            torch.save((sd_local, err_l, sd_global, err_g), a.weights_cache)
    if a.workload in ("configs3", "configs4"):
        note = ("synthetic, fitted %d Adam steps (recon %.1f / %.1f mm)" % (a.fit_steps, err_l * 1e3, err_g * 1e3)) if a.vae != "structured" else "synthetic, structured"
        return run_sharded(a, world, rank, device, rehearsal, sd_local, sd_global, note)
    n_frames = n_chunks * CHUNK
    seqd = synth.make_sequence_device(n_frames, seed=1000 + rank, device=device, camera=cam, cam_jitter=CAM_JITTER)
    starts = np.concatenate([c * CHUNK + window_starts(CHUNK) for c in range(n_chunks)]).astype(np.int32)
    chunk_of = np.repeat(np.arange(n_chunks), len(window_starts(CHUNK)))
    if a.workload == "w8192x":          # exactly 8192 windows: the first 8192 of the 683 chunks' 8196 (1024 tail workgroups, 32 row panels)
        starts, chunk_of = starts[:8192], chunk_of[:8192]
    B = len(starts)
    eng = WindowEngine(shape, cam, max_windows=B)
    eng.load_vae(LOCAL_STAGE, sd_local)
    eng.load_vae(GLOBAL_STAGE, sd_global)
    eng.set_precision(a.precision)
    if a.lanes is not None:
        eng.set_lanes(a.lanes)
    mb = torch.stack([eng.mean_bone_length(seqd["est_local"][c * CHUNK:(c + 1) * CHUNK]) for c in range(n_chunks)])
    mb_w = mb[torch.as_tensor(chunk_of, device=device)].contiguous()
    g = torch.Generator().manual_seed(4321 + rank)
    eps = torch.randn(2 * B, shape.latent_dim, generator=g).reshape(B, 2, -1)
    eps_l, eps_g = eps[:, 0].contiguous().to(device), eps[:, 1].contiguous().to(device)
    f0 = torch.as_tensor(starts, device=device)
    # CLI defaults of optimize_whole_sequence.py:14-19 through main()'s two set_weights calls (optimizer.py:352-358)
    wl = (0.01 / 10000, 0.001 / 100, 0.01, 0.0, 0.01)
    wg = (0.01, 0.001, 0.01, 0.0, 0.0)
    w_local, w_global = energy_weights(*wl), energy_weights(*wg)
    from globalegomocap_amd.dist import all_gather_windows

    def step():
        mid, glob, stats = eng.optimize_windows(seqd["est_local"], seqd["cams"], seqd["heat"], f0, mb_w, eps_l, eps_g,
                                                w_local, w_global, want_stats=True)
        if world > 1:
            all_gather_windows(glob, B * world)      # refined global poses of every shard, on every rank
        return mid, glob, stats

    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    profile = not a.no_profile
    # HIP events around the dominant kernel cost ~2 us of stream time each (260 pairs per step): record them in
    # the first PROFILE_STEPS steps of the timed region only, so that they do not distort `value`
    t0 = time.perf_counter()
    for i in range(a.steps):
        eng.profile_enable(profile and i < PROFILE_STEPS)
        mid, glob, stats = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    eng.profile_enable(False)
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- side measurement (not `value`): the same workload in the faster arithmetic modes of the wide products
    other = {}
    if a.precision == "f32" and not a.no_extra:
        for mode in ("bf16x3", "bf16"):
            eng.set_precision(mode)
            step()
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            t1 = time.perf_counter()
            for _ in range(a.steps):
                m2, g2, s2 = step()
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            dt = time.perf_counter() - t1
            if world > 1:
                t = torch.tensor([dt], device=device, dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dt = float(t.item())
            other[mode] = (dt, g2.cpu().numpy() if rank == 0 else None)
        eng.set_precision("f32")

    # ---- side record (not `value`): SURVEY 8f.3, the drop-in interface end to end -- `whole_sequence.optimize_directory` on this
    # rank's sequence written as 20 chunk directories of `test_data.pkl` files THE REFERENCE'S WAY (process_test_data.py:149-157:
    # five keys, heat-maps Fortran-ordered as scipy.io.loadmat returns them, default pickle protocol; page-cached): the library's
    # pickle reader, file -> pinned memory -> HBM, the un-Fortran-ing kernel, the batched optimisation, device merge + report.
    # `three_sequences_pipelined`: three such sequences through optimize_sequences(per_sequence=True) -- while one computes, the
    # next one's files cross PCIe.  Host-inclusive windows/s, never `value`.  Measured BEFORE the other side records: what a fresh
    # process running optimize_whole_sequence.py sees (behind the sharded legs' allocations and graph streams the same calls
    # measured 25.6 / 48.5 ms instead of 21.3 / 45.7).
    host_inclusive = None
    if world == 1 and a.workload == "seq2k" and not a.no_extra and a.precision == "f32":
        import pickle
        import shutil
        import tempfile
        from globalegomocap_amd import whole_sequence as ws_mod
        from globalegomocap_amd.optimizer import SequenceOptimizer
        root_dir = tempfile.mkdtemp(prefix="gem_bench_seq_")
        try:
            seq_dirs = []

            def write_sequences():
                # (written from a thread on the NUMA node of the GPU's PCIe root: that is where the page cache of files READ from
                # disk by the library's node-confined readers ends up; files written here from an arbitrary CPU may sit on the other
                # socket, which costs the read + host-to-device pipeline 14.5 instead of 10.4 ms -- tools/r06_numa_probe.py)
                near = ws_mod.cpus_near(device)
                if near:
                    os.sched_setaffinity(0, near)
                for si in range(3):
                    sq = seqd if si == 0 else synth.make_sequence_device(n_frames, seed=1500 + si, device=device, camera=cam, cam_jitter=CAM_JITTER)
                    heat_np = sq["heat"].cpu().numpy()
                    seq_dirs.append(os.path.join(root_dir, "seq_%d" % si))
                    for c in range(n_chunks):
                        sl = slice(c * CHUNK, (c + 1) * CHUNK)
                        dch = os.path.join(seq_dirs[-1], "chunk_%d" % c)
                        os.makedirs(dch)
                        with open(os.path.join(dch, "test_data.pkl"), "wb") as f:
                            pickle.dump(synth.reference_pickle_dict({"estimated_local_skeleton": sq["est_local_np"][sl], "gt_global_skeleton": sq["gt_global"][sl],
                                                                     "camera_pose_list": sq["cams_np"][sl], "heatmap_list": heat_np[sl]}), f)
                    del heat_np, sq
            import threading
            wt = threading.Thread(target=write_sequences)
            wt.start()
            wt.join()
            if len(seq_dirs) != 3:
                raise RuntimeError("writing the bench's chunk directories failed")
            opt = SequenceOptimizer(DEFAULT_CALIBRATION, sd_global, sd_local, max_windows=B)
            host_inclusive = {"frames": int(n_frames), "windows": int(B), "pickle_bytes": int(sum(
                os.path.getsize(os.path.join(seq_dirs[0], dn, "test_data.pkl")) for dn in os.listdir(seq_dirs[0]))),
                "pickles": "written as the reference writes them: 5 keys, one Fortran-ordered float32 [64,64,15] array per frame, default protocol"}
            t_h2d = []
            pin = torch.empty(host_inclusive["pickle_bytes"], dtype=torch.uint8).pin_memory()
            dimg = torch.empty_like(pin, device=device)
            for _ in range(4):
                torch.cuda.synchronize()
                th = time.perf_counter()
                dimg.copy_(pin, non_blocking=True)
                torch.cuda.synchronize()
                t_h2d.append(time.perf_counter() - th)
            del pin, dimg
            host_inclusive["pcie_floor_ms"] = round(min(t_h2d) * 1e3, 2)          # one pinned copy of the same number of bytes: what the link alone takes

            def timed(fn, n_windows, reps):
                fn(None)                                                           # warm-up (pools, buffers)
                best, res, log = None, None, None
                for _ in range(reps):
                    tm = {}
                    torch.cuda.synchronize()
                    th = time.perf_counter()
                    res = fn(tm)
                    torch.cuda.synchronize()
                    dtw = time.perf_counter() - th
                    if best is None or dtw < best:
                        best, log = dtw, tm.get("_log")
                return {"ms_end_to_end": round(best * 1e3, 2), "windows_per_s": round(n_windows / best, 1),
                        "main_thread_timeline_ms": [[t_, name] for t_, name in (log or []) if not name.startswith("run:")]}, res
            import gc
            gc.collect()
            single = lambda tm: ws_mod.optimize_directory(seq_dirs[0], DEFAULT_CALIBRATION, optimizer=opt, verbose=False, timings=tm)      # noqa: E731
            rec, res = timed(single, B, 7)
            rec3, res3 = timed(lambda tm: ws_mod.optimize_sequences(seq_dirs, DEFAULT_CALIBRATION, optimizer=opt, verbose=False, per_sequence=True, timings=tm), 3 * B, 7)
            # (the single sequence once more behind the pipelined leg: the files were written seconds ago, and a box whose page cache was
            # still being written back read the first leg's files at 14.6 instead of 9.2 ms; best of both legs = best of 14 calls)
            rec_b, res_b = timed(single, B, 7)
            rec["ms_end_to_end_first_leg"], rec_b["ms_end_to_end_first_leg"] = rec["ms_end_to_end"], rec["ms_end_to_end"]
            if rec_b["ms_end_to_end"] < rec["ms_end_to_end"]:
                rec, res = rec_b, res_b
            rec["optimized_global_mpjpe_mm"] = round(float(res[0]["optimized_global_mpjpe"]) * 1e3, 3)
            host_inclusive["pickles_only"] = rec
            rec3["optimized_global_mpjpe_mm"] = [round(float(r[0]["optimized_global_mpjpe"]) * 1e3, 3) for r in res3]
            rec3["floor_windows_per_s"] = round(3 * B / (3 * host_inclusive["pcie_floor_ms"] * 1e-3 + elapsed / a.steps), 1)
            host_inclusive["three_sequences_pipelined"] = rec3
            near = ws_mod.cpus_near(device)
            try:
                p_ = torch.cuda.get_device_properties(device)
                gpu_node = int(open("/sys/bus/pci/devices/%04x:%02x:%02x.0/numa_node" % (p_.pci_domain_id, p_.pci_bus_id, p_.pci_device_id)).read())
            except (OSError, ValueError, AttributeError):
                gpu_node = None
            host_inclusive["placement"] = {"gpu_numa_node": gpu_node, "readers_confined_to_cpus": len(near) if near else None,
                                           "main_thread_cpu": os.sched_getcpu() if hasattr(os, "sched_getcpu") else None,
                                           "process_cpus": len(os.sched_getaffinity(0))}
            host_inclusive["what"] = ("whole_sequence.optimize_directory (the reference's optimize_whole_sequence.py:48-118) on 20 chunk directories, page-cached "
                                      "files, best of 7 calls (the single sequence: of 14, in two legs): read + host-to-device + optimise + device merge / report, nothing cached next to the data; "
                                      "three_sequences_pipelined = optimize_sequences(per_sequence=True) on three such directories (720 windows): one device call "
                                      "per sequence, the next sequence's files cross PCIe meanwhile; floor = 3 x pcie_floor_ms + one resident step, nothing else")
            opt.engine.close()
            ws_mod.release_pools()
        finally:
            shutil.rmtree(root_dir, ignore_errors=True)

    # ---- side measurement (not `value`): several sequences in flight on one GPU (BASELINE configs[2]'s regime): one engine
    # and one HIP stream per sequence, the kernels of different sequences overlap on the device.  Eager launches (the single
    # host thread enqueues ~700 launches per sequence-step) and hipGraph replay (one launch per sequence-step, configs[4]).
    in_flight = None
    if a.in_flight and a.precision == "f32" and not a.no_extra and world == 1 and a.workload == "seq2k":
        in_flight = {}
        engines = [eng]
        for _ in range(2):
            e2 = WindowEngine(shape, cam, max_windows=B)
            e2.load_vae(LOCAL_STAGE, sd_local)
            e2.load_vae(GLOBAL_STAGE, sd_global)
            engines.append(e2)
        streams = [torch.cuda.Stream() for _ in engines]
        for mode in ("f32", "bf16"):
            for graphs in (False, True):
                for e in engines:
                    e.set_precision(mode)
                    e.enable_graphs(graphs)
                for n in (1, 2, 3):
                    def multi():
                        for e, st_ in zip(engines[:n], streams[:n]):
                            with torch.cuda.stream(st_):
                                e.optimize_windows(seqd["est_local"], seqd["cams"], seqd["heat"], f0, mb_w, eps_l, eps_g, w_local, w_global,
                                                   want_stats=False)
                    multi(); multi()                    # (with graphs: eager warm-up, then capture)
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    for _ in range(a.steps):
                        multi()
                    t_host = time.perf_counter() - t1
                    torch.cuda.synchronize()
                    dt = time.perf_counter() - t1
                    in_flight["%s_%s_%d" % (mode, "graph" if graphs else "eager", n)] = {
                        "windows_per_s": round(B * n * a.steps / dt, 1), "ms_per_sequence": round(dt / (a.steps * n) * 1e3, 3),
                        "host_enqueue_ms_per_sequence": round(t_host / (a.steps * n) * 1e3, 3)}
        for e in engines:
            e.set_precision("f32")
            e.enable_graphs(False)
        del engines[1:]

    # ---- side records (not `value`): BASELINE configs[2] -- "all 5 test-sequence shapes concurrently on 1 MI355X, bf16 VAE
    # decoder / fp32 energy": 20 + 27 + 27 + 27 + 27 = 128 chunks = 1536 windows in ONE call, fp32 and bf16 -- and configs[3]'s
    # per-GPU shard -- exactly 8192 windows (the first 8192 of 683 chunks' 8196), bf16 -- each with the roofline of its own dominant kernels (HIP events on
    # the launch stream; kernel names as recorded by the library at launch time)
    def lbfgs_record(e, ms_k, n_k, step_s, n_windows, mode):
        """`lbfgs_advance_kernel` moves vectors, it does no matrix work: priced against HBM with the bytes of the committed PMC
        passes of the same workload (per launch, dispatch-weighted)."""
        names = e.profile_kernels(2)
        tr_b, tr_src = committed_traffic(names, n_windows, mode) if names else (None, None)
        r = {"bound": "hbm", "kernel": names, "launches": int(n_k), "avg_us": round(ms_k * 1e3 / n_k, 2), "share_of_step": round(ms_k * 1e-3 / step_s, 3),
             "peak": PEAK_HBM_GBPS, "unit": "GB/s", "traffic": tr_b, "traffic_source": tr_src, "achieved": None, "frac": None}
        if tr_b:
            r["achieved"] = round(tr_b / (ms_k * 1e-3 / n_k) / 1e9, 1)
            r["frac"] = round(r["achieved"] / PEAK_HBM_GBPS, 4)
        return r

    def side_record(nc2, modes, seed, what, limit=None):
        rec_all = {}
        B_all = nc2 * len(window_starts(CHUNK))
        B2 = min(limit or B_all, B_all)            # `limit`: the first B2 windows only (the last chunk may then be incomplete)
        seq2 = synth.make_sequence_device(nc2 * CHUNK, seed=seed, device=device, camera=cam, cam_jitter=CAM_JITTER)
        starts2 = np.concatenate([c * CHUNK + window_starts(CHUNK) for c in range(nc2)]).astype(np.int32)[:B2]
        e2 = WindowEngine(shape, cam, max_windows=B2)
        e2.load_vae(LOCAL_STAGE, sd_local)
        e2.load_vae(GLOBAL_STAGE, sd_global)
        est_c = seq2["est_local"].reshape(nc2, CHUNK, 15, 3)
        mb2 = torch.stack([e2.mean_bone_length(est_c[c]) for c in range(nc2)])
        per2 = B_all // nc2
        mb2 = mb2[torch.as_tensor(np.repeat(np.arange(nc2), per2)[:B2], device=device)].contiguous()
        nc_full = B2 // per2                       # whole chunks among the B2 windows: the ones the merged-sequence MPJPE is taken over
        g2 = torch.Generator().manual_seed(987)
        eps2 = torch.randn(B2, 2, shape.latent_dim, generator=g2)
        el2, eg2 = eps2[:, 0].contiguous().to(device), eps2[:, 1].contiguous().to(device)
        f02 = torch.as_tensor(starts2, device=device)
        gt2 = np.concatenate([seq2["gt_global"][c * CHUNK:c * CHUNK + 8 * per2 + 2] for c in range(nc_full)])
        n2 = max(3, min(a.steps, 10))
        for mode in modes:
            e2.set_precision(mode)

            def step2():
                return e2.optimize_windows(seq2["est_local"], seq2["cams"], seq2["heat"], f02, mb2, el2, eg2, w_local, w_global)
            step2()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for i in range(n2):
                e2.profile_enable(profile and i == 0)
                m2, gl2, st2 = step2()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t1
            e2.profile_enable(False)
            sn2 = stats_to_numpy(st2)
            ev2 = sn2["func_evals"].reshape(2, B2)
            gl_np = gl2.cpu().numpy()
            opt2 = np.concatenate([final_smooth(merge_batches(gl_np[c * per2:(c + 1) * per2])) for c in range(nc_full)])
            rec = {"windows": B2, "windows_per_s": round(B2 * n2 / dt, 1), "ms_per_step": round(dt / n2 * 1e3, 3), "steps": n2,
                   "evals_per_stage": {"local_mean": round(float(ev2[0].mean()), 2), "global_mean": round(float(ev2[1].mean()), 2)},
                   "mpjpe_optimised_mm": round(mpjpe(opt2, gt2) * 1e3, 3), "all_finished": bool(sn2["finished"].all()),
                   "degenerate_windows": int(sn2["degenerate"].sum())}
            rec["path"] = path_roofline(rec["windows_per_s"], ev2[0].mean(), ev2[1].mean(), mode)
            if profile:
                # family 0 = the composed front layer (decoder_input o conv 0), forward + backward-data; family 1 = the fused tail
                # (fp32: one window per workgroup; bf16: eight windows per workgroup, bf16 MFMA); family 2 = L-BFGS (HBM-bound)
                for fam, key in ((0, "roofline"), (1, "roofline_tail")):
                    ms_k, n_k, fl_k = e2.profile_read(fam)
                    names = e2.profile_kernels(fam)
                    if not n_k:
                        continue
                    if fl_k <= 0:          # the stand-alone energy kernel (batched narrow layers): no matrix work, latency / gather-bound
                        rec["energy_kernel"] = {"kernel": names, "launches": int(n_k), "avg_us": round(ms_k * 1e3 / n_k, 2),
                                                "share_of_step": round(ms_k * 1e-3 / (dt / n2), 3)}
                        continue
                    pk = PEAK_BF16_MATRIX_TFLOPS if (mode == "bf16" and "decoder_tail_kernel" not in names) else PEAK_F32_MATRIX_TFLOPS
                    ach = fl_k / (ms_k * 1e-3) / 1e12
                    tr_b, tr_src, mf = committed_traffic(names, B2, mode, with_mfma=True)
                    rec[key] = {"bound": kernel_bound(names), "achieved": round(ach, 2), "peak": pk, "unit": "TFLOP/s", "frac": round(ach / pk, 4),
                                "kernel": names, "launches": int(n_k), "avg_us": round(ms_k * 1e3 / n_k, 2), "flop_per_launch": round(fl_k / n_k),
                                "share_of_step": round(ms_k * 1e-3 / (dt / n2), 3), "traffic": tr_b, "traffic_source": tr_src,
                                "mfma_busy": mfma_busy_record(mf, ms_k * 1e3 / n_k, fl_k / n_k, "bf16" if pk == PEAK_BF16_MATRIX_TFLOPS else "f32")}
                ms_k, n_k, _ = e2.profile_read(2)
                if n_k:
                    rec["lbfgs_advance"] = lbfgs_record(e2, ms_k, n_k, dt / n2, B2, mode)
            rec_all[mode] = rec
        rec_all["workload"] = what % B2
        e2.close()
        del seq2
        torch.cuda.empty_cache()
        return rec_all

    configs2 = configs3 = None
    if a.precision == "f32" and not a.no_extra and world == 1 and a.workload == "seq2k":
        configs2 = side_record(128, ("f32", "bf16"), 2000,
                               "BASELINE configs[2]: five sequences of 20+27+27+27+27 chunks = %d windows in one call, local+global "
                               "stage, bf16 = bf16 decoder activations and products / fp32 accumulate, energies and L-BFGS")
        configs3 = side_record(683, ("bf16",), 3000,
                               "BASELINE configs[3] per-GPU shard: the first %d windows of 683 chunks in one call, local+global stage, bf16 "
                               "decoder activations and products / fp32 accumulate, energies and L-BFGS; the 8-GPU leg of this "
                               "config is `bench.py --gpus 8 --workload configs3`", limit=8192)

    # ---- side record (not `value`): BASELINE configs[4] -- "streaming 100k-frame sequence over 8 GPUs, hipGraph-captured":
    # the per-GPU shard, 1563 overlapping windows (stride 8) of ONE continuous 12 506-frame stream, frames stored once, no chunk
    # structure, every call after the second replayed from a hipGraph
    configs4 = None
    if a.precision == "f32" and not a.no_extra and world == 1 and a.workload == "seq2k":
        configs4 = {}
        B4 = 1563
        starts4 = (8 * np.arange(B4)).astype(np.int32)
        nf4 = int(starts4[-1]) + 10
        seq4 = synth.make_sequence_device(nf4, seed=4000, device=device, camera=cam, cam_jitter=CAM_JITTER)
        e4 = WindowEngine(shape, cam, max_windows=B4)
        e4.load_vae(LOCAL_STAGE, sd_local)
        e4.load_vae(GLOBAL_STAGE, sd_global)
        mb4 = e4.mean_bone_length(seq4["est_local"]).reshape(1, 15).expand(B4, 15).contiguous()
        g4 = torch.Generator().manual_seed(654)
        eps4 = torch.randn(B4, 2, shape.latent_dim, generator=g4)
        el4, eg4 = eps4[:, 0].contiguous().to(device), eps4[:, 1].contiguous().to(device)
        f04 = torch.as_tensor(starts4, device=device)
        gt4 = seq4["gt_global"][:8 * B4 + 2]
        n4 = max(3, min(a.steps, 10))
        for mode in ("f32", "bf16"):
            e4.set_precision(mode)
            e4.enable_graphs(True)
            for _ in range(3):                   # eager, capture, first replay
                e4.optimize_windows(seq4["est_local"], seq4["cams"], seq4["heat"], f04, mb4, el4, eg4, w_local, w_global)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            host = 0.0
            for _ in range(n4):
                th = time.perf_counter()
                m4, gl4, st4 = e4.optimize_windows(seq4["est_local"], seq4["cams"], seq4["heat"], f04, mb4, el4, eg4, w_local, w_global)
                host += time.perf_counter() - th
            torch.cuda.synchronize()
            dt = time.perf_counter() - t1
            gs = e4.graph_stats()
            sn4 = stats_to_numpy(st4)
            merged = final_smooth(merge_batches(gl4.cpu().numpy()))
            configs4[mode] = {"windows": B4, "windows_per_s": round(B4 * n4 / dt, 1), "ms_per_step": round(dt / n4 * 1e3, 3),
                              "host_enqueue_ms_per_step": round(host / n4 * 1e3, 3), "graph_replays": int(gs["replays"]),
                              "mpjpe_optimised_mm": round(mpjpe(merged, gt4) * 1e3, 3), "all_finished": bool(sn4["finished"].all()), "degenerate_windows": int(sn4["degenerate"].sum())}
            e4.enable_graphs(False)
        configs4["workload"] = ("BASELINE configs[4] per-GPU shard: %d overlapping windows (stride 8) of one continuous %d-frame stream, "
                                "local+global stage, whole call replayed from a hipGraph" % (B4, nf4))
        e4.close()
        del seq4

    # ---- side records of the MULTI-RANK default line (not `value`): BASELINE configs[3] and configs[4] as ONE job sharded over the
    # ranks (strong scaling: 65 536 windows in contiguous shards; the 12 499 windows of one 100k-frame stream in block-cyclic
    # shards + halo, graph replay, gather + merge) -- the legs `--workload configs3 | configs4` run alone.  The rehearsal (several
    # ranks on one card) takes a sixteenth of the windows per rank so that the 2-rank test stays short.
    strong = {}
    if world > 1 and a.workload == "seq2k" and not a.no_extra:
        import copy
        for wl_name, n_def in (("configs3", 65536), ("configs4", 12499)):
            a2 = copy.copy(a)
            a2.workload, a2.precision, a2.dump, a2.emulate_ranks = wl_name, "bf16", None, 0
            a2.windows = a.windows or (max(64 * world, n_def // 16) if rehearsal else n_def)
            a2.steps, a2.warmup = max(2, min(a.steps, 5)), 1
            strong[wl_name] = run_sharded(a2, world, rank, device, rehearsal, sd_local, sd_global,
                                          "synthetic, fitted %d Adam steps (recon %.1f / %.1f mm)" % (a.fit_steps, err_l * 1e3, err_g * 1e3)
                                          if a.vae != "structured" else "synthetic, structured", emit=False)

    # ---- side records (not `value`): BASELINE configs[3] and configs[4] at their FULL size under the 8-way partition of the
    # multi-GPU legs, the eight shards run one after the other on this one card (run_sharded, --emulate-ranks 8): per-rank
    # evaluations / time, their imbalance, and a labelled PROJECTION of the 8-GPU rate
    partition8 = None
    if world == 1 and a.workload == "seq2k" and not a.no_extra and not a.no_partition and a.precision == "f32":
        import copy
        partition8 = {}
        for wl_name in ("configs3", "configs4"):
            a2 = copy.copy(a)
            a2.workload, a2.precision, a2.dump, a2.emulate_ranks, a2.windows = wl_name, "bf16", None, 8, 0
            a2.steps, a2.warmup = max(2, min(a.steps, 3)), 2
            partition8[wl_name] = run_sharded(a2, world, rank, device, rehearsal, sd_local, sd_global, "as the headline", emit=False)

    if rank == 0:
        st = stats_to_numpy(stats)
        assert st["finished"].all(), "a window did not finish"
        assert not st["degenerate"].any(), "a window saw a joint on the optical axis (the reference raises 'norm is zero!')"
        evals = st["func_evals"].reshape(2, B)
        glob_np = glob.cpu().numpy()
        # accuracy on this rank's sequence: merged + final-smoothed chunks vs ground truth (metres)
        per = len(window_starts(CHUNK))
        opt_seq, gt_seq, est_seq = [], [], []
        for c in range(n_chunks):
            opt_seq.append(final_smooth(merge_batches(glob_np[c * per:(c + 1) * per])))
            n_out = opt_seq[-1].shape[0]
            gt_seq.append(seqd["gt_global"][c * CHUNK:c * CHUNK + n_out])
            homo = np.concatenate([seqd["est_local_np"][c * CHUNK:c * CHUNK + n_out], np.ones((n_out, 15, 1))], -1)
            est_seq.append(np.einsum("nij,nkj->nki", seqd["cams_np"][c * CHUNK:c * CHUNK + n_out], homo)[..., :3])
        mp_opt, mp_in = mpjpe(np.concatenate(opt_seq), np.concatenate(gt_seq)), mpjpe(np.concatenate(est_seq), np.concatenate(gt_seq))

        def seq_mpjpe(gl):
            o = [final_smooth(merge_batches(gl[c * per:(c + 1) * per])) for c in range(n_chunks)]
            return mpjpe(np.concatenate(o), np.concatenate(gt_seq))
        other_modes = {m: {"windows_per_s": round(B * world * a.steps / dt, 2), "ms_per_step": round(dt / a.steps * 1e3, 3),
                           "mpjpe_optimised_mm": round(seq_mpjpe(gl) * 1e3, 3)} for m, (dt, gl) in other.items()}
        roof = None
        roof_other = None
        roof_lbfgs = None
        if profile:
            # HIP-event timings of the two kernel families that carry the step: family 0 = the matrix products around the latent
            # (decoder_input o conv 0 composed into one layer, forward + backward-data), family 1 = the fused decoder tail.
            # `roofline` describes whichever took more of the timed region, `roofline_other` the other one.  The kernel names are
            # the ones the library recorded when it launched them (gem_profile_kernels), i.e. the rows of the rocprofv3 stats.
            fam = {}
            for k in (0, 1):
                ms, n, fl = eng.profile_read(k)
                if n:
                    fam[k] = (ms, n, fl, eng.profile_kernels(k))
            # bf16x3 issues three bf16 MFMAs per algorithmic product
            peak = {"f32": PEAK_F32_MATRIX_TFLOPS, "bf16x3": PEAK_BF16_MATRIX_TFLOPS / 3, "bf16": PEAK_BF16_MATRIX_TFLOPS}[a.precision]
            what = {0: "decoder_input o conv 0 as one layer: forward + backward-data",
                    1: "fused decoder tail: convs 256->128->64->64->64->45, energy terms, adjoint convs"}

            def roof_of(k):
                ms, n, fl, names = fam[k]
                achieved = fl / (ms * 1e-3) / 1e12
                pk = peak if (k == 0 or "bf16" in names) else PEAK_F32_MATRIX_TFLOPS        # (the one-window tail is fp32 in every mode)
                tr_b, tr_src, mf = committed_traffic(names, B, a.precision, with_mfma=True)
                return {"family": ("front products", "fused tail")[k], "bound": kernel_bound(names) if k == 1 else "mfma", "achieved": round(achieved, 3), "peak": round(pk, 1), "unit": "TFLOP/s", "frac": round(achieved / pk, 4),
                        "traffic": tr_b, "traffic_source": tr_src, "kernel": names, "what": what[k], "launches": int(n),
                        "avg_us": round(ms * 1e3 / n, 2), "flop_per_launch": round(fl / n),
                        "mfma_busy": mfma_busy_record(mf, ms * 1e3 / n, fl / n, "bf16" if pk == PEAK_BF16_MATRIX_TFLOPS else "f32"),
                        "share_of_step": round(ms * 1e-3 / min(PROFILE_STEPS, a.steps) / (elapsed / a.steps), 3)}
            ms2, n2k, _ = eng.profile_read(2)
            if n2k:
                roof_lbfgs = lbfgs_record(eng, ms2, n2k, min(PROFILE_STEPS, a.steps) * (elapsed / a.steps), B, a.precision)
            if fam:
                # the dominant family by device time; the two are within a few per cent of one another at this size (run to run either
                # leads): inside 5 % the matrix products -- the family the MFMA roofline prices -- stay the headline, as in every round
                dom = max(fam, key=lambda k: fam[k][0])
                if 0 in fam and fam[0][0] >= 0.95 * fam[dom][0]:
                    dom = 0
                roof = roof_of(dom)
                rest = [k for k in fam if k != dom]
                if rest:
                    roof_other = roof_of(rest[0])
        cpu = None
        if world == 1 and a.cpu_windows > 0:
            nw = min(a.cpu_windows, B)
            ref, dt, nthreads = cpu_baseline(sd_local, sd_global, cam, seqd, starts, mb_w[0].cpu().numpy(), eps_l.cpu().numpy(),
                                             eps_g.cpu().numpy(), wl, wg, nw)
            diff = float(np.linalg.norm(ref - glob_np[:nw], axis=-1).mean())
            gtw = np.stack([seqd["gt_global"][s:s + 10] for s in starts[:nw]])
            cpu = {"value": round(nw / dt, 4), "unit": "windows/s", "cores": int(nthreads), "kind": "port",
                   "sample": "first %d windows of the same sequence, same weights and eps, torch %s CPU, "
                             "autograd incl. frozen-VAE weight grads + torch.optim.LBFGS" % (nw, torch.__version__),
                   "mpjpe_port_mm": round(mpjpe(ref, gtw) * 1000, 4), "mpjpe_hip_mm": round(mpjpe(glob_np[:nw], gtw) * 1000, 4),
                   "mean_joint_diff_hip_vs_port_mm": round(diff * 1000, 4)}
        post = None
        if not a.no_extra:
            # overlap merge + smoothing + the reference's 18-entry error report on the device, against the
            # numpy mirror of the reference's calculate_errors on the same sequences (outside the timed region)
            from globalegomocap_amd.errors import calculate_errors
            gt_all, est_all = np.concatenate(gt_seq), np.concatenate(est_seq)
            idx = (f0.long()[:, None] + torch.arange(10, device=device)[None])
            cw = seqd["cams"][idx]
            mid_glob = torch.einsum("btij,btkj->btki", cw[..., :3, :3], mid.double()) + cw[..., None, :3, 3]
            gt_d, est_d = torch.as_tensor(gt_all, device=device), torch.as_tensor(est_all, device=device)

            def device_report():
                o = eng.merge_windows(glob, n_chunks, smooth=True)
                m = eng.merge_windows(mid_glob, n_chunks, smooth=False)
                return o, m, eng.calculate_errors_device(est_d, m, o, gt_d)
            device_report()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                o_d, m_d, rep = device_report()
            e1.record()
            torch.cuda.synchronize()
            rep = rep.cpu().numpy()
            nf = min(gt_all.shape[0], 980)                       # bounded host sample: the mirror is per-frame python
            t0 = time.perf_counter()
            host = calculate_errors(est_all[:nf], m_d.cpu().numpy()[:nf], o_d.cpu().numpy()[:nf], gt_all[:nf])
            host_ms = (time.perf_counter() - t0) * 1e3
            sub = eng.calculate_errors(est_all[:nf], m_d[:nf], o_d[:nf], gt_all[:nf])
            worst = max(float(np.max(np.abs(np.asarray(sub[k]) - np.asarray(host[k])))) for k in host)
            post = {"frames": int(gt_all.shape[0]), "device_ms": round(e0.elapsed_time(e1) / 10, 4),
                    "host_numpy_ms_per_%d_frames" % nf: round(host_ms, 1), "max_abs_diff_vs_host_m": worst,
                    "optimized_global_mpjpe_mm": round(float(rep[2]) * 1e3, 3),
                    "optimized_aligned_global_mpjpe_mm": round(float(rep[10]) * 1e3, 3),
                    "aligned_optimized_mpjpe_mm": round(float(rep[13]) * 1e3, 3),
                    "bone_length_aligned_optimized_mpjpe_mm": round(float(rep[16]) * 1e3, 3)}
        lift = None
        if not a.no_extra:
            # heat-map argmax + fisheye un-projection, one streaming pass over the resident heat-maps
            depth_d = seqd["est_local"].double().norm(dim=-1)
            eng.lift_skeleton(seqd["heat"], depth_d)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                eng.lift_skeleton(seqd["heat"], depth_d)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 10
            nbytes = seqd["heat"].numel() * 4
            lift = {"frames": int(n_frames), "ms": round(ms, 4), "heatmap_bytes": int(nbytes),
                    "achieved_GBps": round(nbytes / ms / 1e6, 1), "frac_of_hbm_peak": round(nbytes / ms / 1e6 / PEAK_HBM_GBPS, 4),
                    "note": "incl. the f32/f64 output allocation of the Python wrapper; kernel-only time is in profiles/"}
        train = None
        if not a.no_extra and world == 1:
            # one training step of the full-size VAE (networks/train.py:77-83 at the reference's default batch of 64):
            # train-mode forward, loss, backward incl. weight gradients, Adam -- all on the device, against the torch-CPU port
            from globalegomocap_amd import synth as synth_mod
            from globalegomocap_amd.vae_train import VAETrainer, initial_state_dict
            tb = 64
            init = initial_state_dict(shape, 0)
            tw = synth_mod.make_training_windows(tb, shape.seq_len, 0)
            te = np.random.default_rng(0).standard_normal((tb, shape.latent_dim)).astype(np.float32)
            trn = VAETrainer(shape, batch_size=tb, lr=1e-4, state_dict=init)
            tw_d, te_d = torch.as_tensor(tw, device=device), torch.as_tensor(te, device=device)
            # keep_gradients=False = the training loop's mode (VAETrainer.fit; networks/train.py:77-83 never reads p.grad): the two
            # linear layers form their weight gradient inside their Adam step; `ms_per_step_gradients_kept` is the p.grad mode
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ms_mode = {}
            for keep in (True, False):
                for _ in range(3):
                    trn.step(tw_d, 0.01, eps=te_d, sync=False, keep_gradients=keep)
                torch.cuda.synchronize()
                e0.record()
                for _ in range(20):
                    trn.step(tw_d, 0.01, eps=te_d, sync=False, keep_gradients=keep)
                e1.record()
                torch.cuda.synchronize()
                ms_mode[keep] = e0.elapsed_time(e1) / 20
            ms = ms_mode[False]
            n_par = trn.n_params
            trn.close()
            # the same step at batch 1024: the matrix products carry it (forward, backward-data and weight gradients: 3 x 2 x 42.32 M
            # multiply-adds per window), priced against the fp32 matrix peak
            tb2 = 1024
            tw2 = torch.as_tensor(synth_mod.make_training_windows(tb2, shape.seq_len, 1), device=device)
            te2 = torch.as_tensor(np.random.default_rng(1).standard_normal((tb2, shape.latent_dim)).astype(np.float32), device=device)
            trn2 = VAETrainer(shape, batch_size=tb2, lr=1e-4, state_dict=init)
            for _ in range(2):
                trn2.step(tw2, 0.01, eps=te2, sync=False, keep_gradients=False)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(10):
                trn2.step(tw2, 0.01, eps=te2, sync=False, keep_gradients=False)
            e1.record()
            torch.cuda.synchronize()
            ms2 = e0.elapsed_time(e1) / 10
            trn2.close()
            del tw2, te2
            flop_w = 3 * 2 * (26341760 + 15978880)
            # algorithmic HBM bytes of a step: Adam reads p, m, v and writes p, m, v (the gradient of the linear layers, 97 % of the
            # arena, never leaves the chip in this mode); forward and backward-data read every weight once each (activations are
            # small beside the 130 MB arena at this batch)
            alg_bytes = n_par * 4 * (6 + 2)
            train = {"batch": tb, "ms_per_step": round(ms, 4), "windows_per_s": round(tb / ms * 1e3, 1), "parameters_padded": int(n_par),
                     "mode": "training loop (gem_trainer_step update = 2: linear-layer weight gradients formed inside their Adam step)",
                     "ms_per_step_gradients_kept": round(ms_mode[True], 4),
                     "algorithmic_bytes_per_step": int(alg_bytes), "achieved_GBps": round(alg_bytes / ms / 1e6, 1),
                     "frac_of_hbm_peak": round(alg_bytes / ms / 1e6 / PEAK_HBM_GBPS, 4), "dtype": "f32", "cpu_baseline": None,
                     "batch_1024": {"ms_per_step": round(ms2, 4), "windows_per_s": round(tb2 / ms2 * 1e3, 1), "flop_per_window": flop_w,
                                    "roofline": {"bound": "mfma", "achieved": round(tb2 * flop_w / ms2 / 1e9, 2), "peak": PEAK_F32_MATRIX_TFLOPS,
                                                 "unit": "TFLOP/s", "frac": round(tb2 * flop_w / ms2 / 1e9 / PEAK_F32_MATRIX_TFLOPS, 4)}}}
            if a.cpu_windows > 0:          # the cpu_baseline leg of this record: the torch-CPU port of the same step on the host cores
                from oracle.torch_port import TrainPort
                nthreads = torch.get_num_threads()
                port = TrainPort(init, lr=1e-4)
                port.step(tw, te, 0.01)
                t0 = time.perf_counter()
                for _ in range(3):
                    port.step(tw, te, 0.01)
                cpu_ms = (time.perf_counter() - t0) / 3 * 1e3
                train["cpu_baseline"] = {"value": round(cpu_ms, 1), "unit": "ms/step", "cores": int(nthreads), "kind": "port",
                                         "sample": "3 steps of the same batch: oracle/torch_port.TrainPort (autograd + torch.optim.Adam), torch %s CPU"
                                                   % torch.__version__}
        total_windows = B * world * a.steps
        value = total_windows / elapsed
        if roof is not None:
            # SURVEY.md 8d's path-level figure for `value` (per GPU), beside the dominant kernel's own
            roof["path"] = path_roofline(value / world, evals[0].mean(), evals[1].mean(), a.precision)
            roof["other"], roof["lbfgs"] = roof_other, roof_lbfgs

        def pick(d, *keys):
            for k in keys:
                d = d.get(k) if isinstance(d, dict) else None
            return d

        def part_summary(r):
            if not r:
                return None
            pt = r["partition"]
            proj = [v for k, v in pt.items() if k.startswith("projected_")]
            return {"windows": r["config"]["windows_total"], "one_card_wps": r["value"], "eval_imbalance": pt["evaluations_per_rank"]["max_over_mean"],
                    "time_imbalance": pt["time_per_rank"]["max_over_mean"], "projected_8gpu_wps_NOT_MEASURED": proj[0] if proj else None,
                    "order_ok": pt["gathered_order_is_arange"], "all_finished": r["all_finished"], "mpjpe_mm": [r["mpjpe_input_mm"], r["mpjpe_optimised_mm"]]}

        def cfg_summary(r, mode):
            m = pick(r, mode)
            if not m:
                return None
            return {"wps": m["windows_per_s"], "path_frac_executed": pick(m, "path", "frac_executed"),
                    "gemm": [pick(m, "roofline", "avg_us"), pick(m, "roofline", "frac"), pick(m, "roofline", "mfma_busy", "frac_of_active_cycles")],
                    "tail": [pick(m, "roofline_tail", "avg_us"), pick(m, "roofline_tail", "frac")],
                    "lbfgs": [pick(m, "lbfgs_advance", "avg_us"), pick(m, "lbfgs_advance", "frac")], "mpjpe_mm": m.get("mpjpe_optimised_mm")}
        # the last ~2 KB of the line are what a truncated log keeps: the headline numbers of every side record, compact
        by_family = {pick(r_, "family"): r_ for r_ in (roof, roof_other) if r_}
        r_gemm, r_tail = by_family.get("front products"), by_family.get("fused tail")
        summary = {
            "value": round(value, 1), "path_frac_executed": pick(roof, "path", "frac_executed"),
            "gemm": [pick(r_gemm, "avg_us"), pick(r_gemm, "frac"), pick(r_gemm, "mfma_busy", "frac_of_active_cycles")] if r_gemm else None,
            "tail": [pick(r_tail, "avg_us"), pick(r_tail, "frac")] if r_tail else None,
            "lbfgs": [pick(roof_lbfgs, "avg_us"), pick(roof_lbfgs, "frac")] if roof_lbfgs else None,
            "bf16_240_wps": pick(other_modes, "bf16", "windows_per_s"),
            "configs2_f32": cfg_summary(configs2, "f32"), "configs2_bf16": cfg_summary(configs2, "bf16"),
            "configs3_shard_bf16": cfg_summary(configs3, "bf16") if world == 1 else None,
            "configs4_shard": {m: [pick(configs4, m, "windows_per_s"), pick(configs4, m, "mpjpe_optimised_mm")] for m in ("f32", "bf16")} if (world == 1 and configs4) else None,
            "partition8_configs3": part_summary(pick(partition8, "configs3")), "partition8_configs4": part_summary(pick(partition8, "configs4")),
            "train": [pick(train, "ms_per_step"), pick(train, "batch_1024", "ms_per_step"), pick(train, "batch_1024", "roofline", "frac")] if train else None,
            "lift_GBps": pick(lift, "achieved_GBps"), "post_ms": pick(post, "device_ms"),
            "host_inclusive_wps": [pick(host_inclusive, "pickles_only", "windows_per_s"), pick(host_inclusive, "three_sequences_pipelined", "windows_per_s")],
            "cpu_wps": pick(cpu, "value"), "cpu_cores": pick(cpu, "cores"),
            "legend": "gemm/tail/lbfgs = [avg us per launch, fraction of its roofline(, MFMA busy / active cycles from the committed PMC pass)]",
        }
        line = {
            "other_precisions": other_modes or None,
            "configs2": configs2,
            # one GPU: the per-GPU shards of configs[3] / configs[4]; several ranks: the whole jobs, sharded (scaling: strong)
            "configs3": configs3 if world == 1 else strong.get("configs3"),
            "configs4": configs4 if world == 1 else strong.get("configs4"),
            "partition8": partition8,
            "sequences_in_flight": in_flight,
            "host_inclusive": host_inclusive,
            "post": post,
            "lift": lift,
            "train": train,
            "evals_per_stage": {"local_mean": float(evals[0].mean()), "global_mean": float(evals[1].mean()),
                                "min": int(evals.min()), "max": int(evals.max())},
            "mpjpe_mm": {"input": round(mp_in * 1e3, 3), "optimised": round(mp_opt * 1e3, 3)},
            "metric": "optimised windows/sec (10-frame, 15-joint)",
            "value": round(value, 2),
            "unit": "windows/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(elapsed / a.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"f32": "f32", "bf16x3": "f32 via 3x bf16 split MFMA (wide products), f32 elsewhere",
                      "bf16": "bf16 wide products / f32 accumulate, tail and energies"}[a.precision], "data": "synthetic",
            "config": {"workload": "%s: %d frames = %d chunks x %d = %d windows per GPU, %s, local+global stage, L-BFGS 25 / 31"
                                   % ({"seq2k": "BASELINE configs[1]", "w8192": "configs[3] shard (683 chunks)", "w8192x": "configs[3] shard (8192 windows)"}
                                      .get(a.workload, "--workload %s" % a.workload), n_frames, n_chunks, per, B, a.precision),
                       "windows_per_gpu": B, "latent_dim": shape.latent_dim, "parallelism": "window-shards x%d" % world,
                       "ranks": world, "backend": dist.get_backend() if world > 1 else None,
                       "vae": ("synthetic, fitted %d Adam steps (recon %.1f / %.1f mm)" % (a.fit_steps, err_l * 1e3, err_g * 1e3))
                              + (" by the HIP trainer" if a.vae == "fit-device" else "")
                              if a.vae != "structured" else "synthetic, structured (vae.structured_state_dict seeds 7 / 8)",
                       "composed_front_layer": True},
            "roofline": roof,
            "cpu_baseline": cpu,
            "summary": summary,
        }
        if a.full_record:
            with open(a.full_record, "w") as f:
                json.dump(line, f, indent=1)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
