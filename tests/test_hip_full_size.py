"""GPU parity tests at the reference's REAL size (D = 2048) and at every BASELINE.json workload.

* the reference's own full-size `main()` run (tests/golden/pipeline_full.npz, made by oracle/make_golden_full.py from the
  unmodified reference): every stage call in isolation, then the exact call bench.py times (`optimize_windows`: both stages,
  jittered cameras, CLI weights) and the `main()` mirror;
* BASELINE configs[2] (all five test-sequence shapes = 128 chunks = 1536 windows in ONE call, bf16 decoder), configs[3]
  (the per-GPU shard of 64k windows = 8192 windows, bf16) and configs[4] (the per-GPU shard of a 100k-frame stream =
  1563 overlapping windows of one continuous sequence), each through size-independent properties plus an oracle spot check.
"""
import json
import os
import pickle

import numpy as np
import pytest

from globalegomocap_amd import synth, vae as vae_schema
from globalegomocap_amd.camera import FisheyeCamera, DEFAULT_CALIBRATION
from globalegomocap_amd.sequence import window_starts, merge_batches, final_smooth
from oracle import np_oracle as O
from helpers import FULL, oracle_camera, full_golden_case

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = dict(vae_weight=0.0, gmm_weight=0.0, smoothness_weight=0.001, bone_length_weight=0.01, weight_3d=0.01, reproj_weight=0.01)
W_LOCAL = (0.01 / 10000, 0.001 / 100, 0.01, 0.0, 0.01)       # optimizer.py:355-358 at the CLI defaults
W_GLOBAL = (0.01, 0.001, 0.01, 0.0, 0.0)                      # optimizer.py:352-353


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("gpu tests need a HIP device")
    return torch


@pytest.fixture(scope="module")
def full_vaes():
    """The two structured full-size VAEs of the golden run (regenerated from their seeds, SHA-256 checked there)."""
    g = np.load(os.path.join(ROOT, "tests", "golden", "pipeline_full.npz"))
    return full_golden_case(g)


@pytest.fixture(scope="module")
def friendly_vaes():
    """The same two structured full-size VAEs in the bf16-friendly gauge (vae.structured_state_dict(signal_offset=1)): the signal
    channels carry 1 + u instead of 3 + u, so that bf16's 2^-8 relative resolution is ~1 mm of pose instead of ~5 mm.  Same
    function in exact arithmetic; regenerated from the seeds like the others.  Used by the BASELINE configs[2..4] tests, whose
    bf16 runs are pinned to the fp32 CPU oracle."""
    sd_l = vae_schema.structured_state_dict(FULL, 7, feature_offset=0.0, signal_offset=1.0)
    sd_g = vae_schema.structured_state_dict(FULL, 8, feature_offset=3.0, signal_offset=1.0)
    return sd_l, sd_g


def _engine(max_windows, sd_l, sd_g, precision="f32", calibration=DEFAULT_CALIBRATION):
    from globalegomocap_amd.engine import WindowEngine, LOCAL_STAGE, GLOBAL_STAGE
    eng = WindowEngine(FULL, FisheyeCamera.from_json(calibration), max_windows=max_windows)
    eng.load_vae(LOCAL_STAGE, sd_l)
    eng.load_vae(GLOBAL_STAGE, sd_g)
    eng.set_precision(precision)
    return eng


def _ew(w):
    from globalegomocap_amd.engine import energy_weights
    return energy_weights(*w)


def _report(name, payload):
    """Numbers worth keeping (deviation histograms) go to gpurun_out/ so that DESIGN.md can quote a measured run."""
    d = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, name), "w") as f:
            json.dump(payload, f, indent=1)
    except OSError:
        pass


# ------------------------------------------------------------------------------------------------------------------
# the reference's full-size run
# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["pipeline_full", "pipeline_full_allterms", "pipeline_full_altcam", "pipeline_full_nlglobal"])
def test_full_size_stages_against_reference_golden(torch_cuda, golden, name):
    """The 24 stage calls of the reference's main() at D = 2048, each from the reference's OWN stage input: closure traces,
    (n_iter, func_evals) and result poses.  Global stages (smooth energy) are pinned to rounding; local stages up to the
    kinks of the bilinear heat-map sampling (see tests/test_oracle_golden.py for the same statement about the CPU oracle)."""
    from globalegomocap_amd.engine import stats_to_numpy
    from helpers import FULL_GOLDENS, golden_calibration
    g = golden(name)
    lim = FULL_GOLDENS[name]
    data, sd_l, sd_g, w_l, w_g = full_golden_case(g)
    eng = _engine(12, sd_l, sd_g, calibration=golden_calibration(g))
    mb = eng.mean_bone_length(data["estimated_local_skeleton"].astype(np.float32))
    starts = window_starts(100)
    heat = data["heatmap_list"]
    rows_report = []
    local_diff = []
    marginal_exits = []
    for st, w in ((0, w_l), (1, w_g)):
        rows = np.arange(st, 24, 2)
        out, stats = eng.optimize_stage(st, g["stage_in"][rows], mb, g["eps"][rows], _ew(w), heat, starts)
        tr = eng.read_trace(12)
        sn = stats_to_numpy(stats)
        out = out.cpu().numpy()
        assert (sn["status"] == 1).all()
        for k, row in enumerate(rows):
            ref_tr = g["trace"][row]
            n_ref = int(g["func_evals"][row])
            n_hip = int(np.isfinite(tr[k]).sum())
            assert n_hip == sn["func_evals"][k], (row, n_hip, sn["func_evals"][k])      # a window's evaluations are rounds 0..evals-1
            d = np.linalg.norm(out[k] - g["stage_out"][row], axis=-1)
            rows_report.append({"row": int(row), "stage": "global" if st else "local", "evals_hip": int(sn["func_evals"][k]),
                                "evals_ref": n_ref, "n_iter_hip": int(sn["n_iter"][k]), "n_iter_ref": int(g["n_iter"][row]),
                                "final_loss_hip": float(sn["final_loss"][k]), "final_loss_ref": float(np.nanmin(ref_tr)),
                                "pose_diff_mean_mm": float(d.mean() * 1e3), "pose_diff_max_mm": float(d.max() * 1e3)})
            # (global-stage energies are sums of squared few-mm residuals: decoded poses that differ by 2e-6 m -- the fp32
            # summation order of the decoder -- move them by a few 1e-4 relative; local-stage energies are O(1))
            np.testing.assert_allclose(tr[k, :4], ref_tr[:4], rtol=5e-4 if st else 2e-4, atol=1e-9, err_msg="row %d" % row)
            if st and (int(sn["func_evals"][k]) != n_ref or int(sn["n_iter"][k]) != int(g["n_iter"][row])):
                # LBFGS.step leaves when |loss - prev_loss| < tolerance_change = 1e-6; the global-stage energies are ~6e-4 and
                # move by about 1e-6 per iteration near the end, so on a window whose last decrease lands within rounding of that
                # threshold the two runs stop an iteration apart -- at energies that agree to the threshold itself.  At most one
                # such window per run, and only with final energies within tolerance_change of each other.
                marginal_exits.append(rows_report[-1])
                assert abs(sn["final_loss"][k] - np.nanmin(ref_tr)) < 1e-6 and abs(int(sn["func_evals"][k]) - n_ref) <= 3, rows_report[-1]
                assert d.mean() < 0.5e-3, rows_report[-1]
                continue
            assert abs(int(sn["func_evals"][k]) - n_ref) <= 1, rows_report[-1]
            assert abs(int(sn["n_iter"][k]) - int(g["n_iter"][row])) <= (1 if st else lim["local_iters"]), rows_report[-1]
            if st:
                assert int(sn["func_evals"][k]) == n_ref and int(sn["n_iter"][k]) == int(g["n_iter"][row]), rows_report[-1]
                np.testing.assert_allclose(tr[k, :n_ref], ref_tr[:n_ref], rtol=1e-3, atol=1e-9, err_msg="row %d" % row)
                assert abs(sn["final_loss"][k] - np.nanmin(ref_tr)) <= lim.get("global_loss", 1e-4) * abs(np.nanmin(ref_tr)), rows_report[-1]
                assert d.mean() < 0.05e-3 and d.max() < lim["global_max"], rows_report[-1]
            else:
                assert abs(sn["final_loss"][k] - np.nanmin(ref_tr)) <= lim["local_loss"] * abs(np.nanmin(ref_tr)), rows_report[-1]
                assert d.mean() < lim["local_mean"], rows_report[-1]
                local_diff.append(d.mean())
    _report("full_size_stage_deviation.json" if name == "pipeline_full" else "full_size_stage_deviation_%s.json" % name, rows_report)
    print("per-stage deviation from the reference (mm, mean over the window's 150 joints):")
    for r in rows_report:
        print("  row %(row)2d %(stage)-6s evals %(evals_hip)d/%(evals_ref)d  n_iter %(n_iter_hip)d/%(n_iter_ref)d  "
              "loss %(final_loss_hip).7e/%(final_loss_ref).7e  diff %(pose_diff_mean_mm).4f (max %(pose_diff_max_mm).4f)" % r)
    assert np.median(local_diff) < lim["local_median"], np.sort(local_diff)
    assert len(marginal_exits) <= 1, marginal_exits
    eng.close()


@pytest.mark.parametrize("name", ["pipeline_full_allterms", "pipeline_full_altcam", "pipeline_full_nlglobal"])
def test_full_size_chained_call_against_the_other_reference_runs(torch_cuda, golden, name):
    """The chained call (`optimize_windows`: local stage -> fp64 relative-global transform -> global stage -> global pose, the
    local RESULT feeding the global stage as in the reference's main()) on the other three reference runs -- every energy term
    on / the 14-coefficient calibration / a non-linear global VAE: merged [98,15,3] output and MPJPE against the reference's
    own main() output (`opt_smooth`, `err_smooth/optimized_global_mpjpe`)."""
    import torch
    from globalegomocap_amd.engine import stats_to_numpy
    from helpers import golden_calibration
    g = golden(name)
    data, sd_l, sd_g, w_l, w_g = full_golden_case(g)
    eng = _engine(12, sd_l, sd_g, calibration=golden_calibration(g))
    dev = eng.device
    est = torch.as_tensor(data["estimated_local_skeleton"], dtype=torch.float32, device=dev).contiguous()
    cams = torch.as_tensor(data["camera_pose_list"], dtype=torch.float64, device=dev).contiguous()
    heat = torch.as_tensor(data["heatmap_list"], dtype=torch.float32, device=dev).contiguous()
    f0 = torch.as_tensor(window_starts(100), dtype=torch.int32, device=dev)
    mb = eng.mean_bone_length(est).reshape(1, 15).expand(12, 15).contiguous()
    eps = torch.as_tensor(g["eps"]).reshape(12, 2, 2048)
    mid, glob, stats = eng.optimize_windows(est, cams, heat, f0, mb, eps[:, 0].contiguous().to(dev), eps[:, 1].contiguous().to(dev),
                                            _ew(w_l), _ew(w_g))
    sn = stats_to_numpy(stats)
    assert (sn["status"] == 1).all()
    ev = sn["func_evals"].reshape(2, 12)
    ref_ev = np.stack([g["func_evals"][0::2], g["func_evals"][1::2]])
    got_mid = merge_batches(mid.cpu().numpy())
    got_opt = final_smooth(merge_batches(glob.cpu().numpy()))
    d_mid = np.linalg.norm(got_mid - g["mid_local_smooth"], axis=-1).mean()
    d_opt = np.linalg.norm(got_opt - g["opt_smooth"], axis=-1).mean()
    mp_hip = np.linalg.norm(got_opt - g["gt_smooth"], axis=-1).mean()
    mp_ref = float(g["err_smooth/optimized_global_mpjpe"])
    print("%s: chained call vs reference main(): merged mid diff %.4f mm, merged optimised diff %.4f mm, MPJPE %.4f vs %.4f mm"
          % (name, d_mid * 1e3, d_opt * 1e3, mp_hip * 1e3, mp_ref * 1e3))
    _report("full_size_chained_call_%s.json" % name, {"mid_diff_mm": float(d_mid * 1e3), "opt_diff_mm": float(d_opt * 1e3),
                                                      "mpjpe_hip_mm": float(mp_hip * 1e3), "mpjpe_ref_mm": float(mp_ref * 1e3),
                                                      "evals_hip": ev.tolist(), "evals_ref": ref_ev.tolist()})
    assert np.abs(ev - ref_ev).max() <= 4, (ev, ref_ev)
    assert d_opt < 0.5e-3 and d_mid < 1.0e-3, (d_mid, d_opt)
    assert abs(mp_hip - mp_ref) < 0.5e-3, (mp_hip, mp_ref)          # north_star's tolerance
    eng.close()


@pytest.mark.parametrize("copies", [20, 19])
def test_full_size_stages_in_a_one_sequence_batch_against_reference_golden(torch_cuda, golden, full_vaes, copies):
    """The same 24 reference stage calls, but batched the way BASELINE configs[1] batches them: 12 windows x 20 (or 19: a
    ragged last row tile) copies = 240 (228) windows per stage call, the size at which the decoder_input products run in
    the few-rows kernel (csrc/gemm_rows.h) and the rows in use shrink on the device while the stage runs.  Every copy must
    reproduce the reference like the 12-window call does -- evaluation counts, closure prefix, final energy, poses -- and
    all copies of a window must agree bitwise (they sit in different row tiles / row blocks)."""
    import torch
    from globalegomocap_amd.engine import stats_to_numpy
    g = golden("pipeline_full")
    data, sd_l, sd_g, w_l, w_g = full_vaes
    B = 12 * copies
    eng = _engine(B, sd_l, sd_g)
    mb = eng.mean_bone_length(data["estimated_local_skeleton"].astype(np.float32))
    starts = np.tile(window_starts(100), copies).astype(np.int32)
    heat = data["heatmap_list"]
    local_diff = []
    for st, w in ((0, w_l), (1, w_g)):
        rows = np.arange(st, 24, 2)
        out, stats = eng.optimize_stage(st, np.tile(g["stage_in"][rows], (copies, 1, 1, 1)), mb, np.tile(g["eps"][rows], (copies, 1)),
                                        _ew(w), heat, starts)
        tr = eng.read_trace(B)
        sn = stats_to_numpy(stats)
        assert (sn["status"] == 1).all()
        o = out.reshape(copies, 12, 10, 15, 3)
        s_ = stats.reshape(copies, 12, -1)
        for c in range(1, copies):
            assert torch.equal(o[0], o[c]) and torch.equal(s_[0], s_[c]), (st, c)
        out = out.cpu().numpy()
        for k, row in enumerate(rows):
            ref_tr = g["trace"][row]
            n_ref = int(g["func_evals"][row])
            d = np.linalg.norm(out[k] - g["stage_out"][row], axis=-1)
            info = (int(row), int(sn["func_evals"][k]), n_ref, float(sn["final_loss"][k]), float(np.nanmin(ref_tr)), float(d.mean() * 1e3))
            np.testing.assert_allclose(tr[k, :4], ref_tr[:4], rtol=5e-4 if st else 2e-4, atol=1e-9, err_msg="row %d" % row)
            assert abs(int(sn["func_evals"][k]) - n_ref) <= 1 and abs(int(sn["n_iter"][k]) - int(g["n_iter"][row])) <= 1, info
            if st:
                assert int(sn["func_evals"][k]) == n_ref and int(sn["n_iter"][k]) == int(g["n_iter"][row]), info
                assert abs(sn["final_loss"][k] - np.nanmin(ref_tr)) <= 1e-4 * abs(np.nanmin(ref_tr)), info
                assert d.mean() < 0.05e-3 and d.max() < 0.2e-3, info
            else:
                assert abs(sn["final_loss"][k] - np.nanmin(ref_tr)) <= 2e-3 * abs(np.nanmin(ref_tr)), info
                assert d.mean() < 2e-3, info
                local_diff.append(d.mean())
    print("B = %d: local-stage pose deviation from the reference (mm): %s" % (B, np.round(np.sort(local_diff) * 1e3, 4)))
    assert np.median(local_diff) < 0.05e-3, np.sort(local_diff)
    eng.close()


def test_full_size_bench_call_against_reference_golden(torch_cuda, golden, full_vaes, tmp_path):
    """The call bench.py times -- WindowEngine.optimize_windows: local stage, fp64 relative-global transform, global stage,
    back to global, all 12 windows of a jittered-camera chunk at once -- and the main() mirror around it, against the
    reference's main() on the same pickle, weights and noise."""
    import torch
    from globalegomocap_amd import optimizer as gopt
    from globalegomocap_amd.engine import stats_to_numpy
    g = golden("pipeline_full")
    data, sd_l, sd_g, w_l, w_g = full_vaes
    eng = _engine(12, sd_l, sd_g)
    dev = eng.device
    est = torch.as_tensor(data["estimated_local_skeleton"], dtype=torch.float32, device=dev).contiguous()
    cams = torch.as_tensor(data["camera_pose_list"], dtype=torch.float64, device=dev).contiguous()
    heat = torch.as_tensor(data["heatmap_list"], dtype=torch.float32, device=dev).contiguous()
    starts = window_starts(100)
    f0 = torch.as_tensor(starts, dtype=torch.int32, device=dev)
    mb = eng.mean_bone_length(est).reshape(1, 15).expand(12, 15).contiguous()
    eps = torch.as_tensor(g["eps"]).reshape(12, 2, 2048)
    mid, glob, stats = eng.optimize_windows(est, cams, heat, f0, mb, eps[:, 0].contiguous().to(dev), eps[:, 1].contiguous().to(dev),
                                            _ew(w_l), _ew(w_g))
    sn = stats_to_numpy(stats)
    assert (sn["status"] == 1).all()
    ev = sn["func_evals"].reshape(2, 12)
    ref_ev = np.stack([g["func_evals"][0::2], g["func_evals"][1::2]])
    assert np.abs(ev - ref_ev).max() <= 3, (ev, ref_ev)
    assert (np.abs(ev - ref_ev) <= 1).sum() >= 20, (ev, ref_ev)
    got_mid = merge_batches(mid.cpu().numpy())
    got_opt = final_smooth(merge_batches(glob.cpu().numpy()))
    d_mid = np.linalg.norm(got_mid - g["mid_local_smooth"], axis=-1).mean()
    d_opt = np.linalg.norm(got_opt - g["opt_smooth"], axis=-1).mean()
    mp_hip = np.linalg.norm(got_opt - g["gt_smooth"], axis=-1).mean()
    mp_ref = float(g["err_smooth/optimized_global_mpjpe"])
    print("bench call vs reference main(): merged mid diff %.4f mm, merged optimised diff %.4f mm, MPJPE %.4f vs %.4f mm"
          % (d_mid * 1e3, d_opt * 1e3, mp_hip * 1e3, mp_ref * 1e3))
    _report("full_size_bench_call.json", {"mid_diff_mm": float(d_mid * 1e3), "opt_diff_mm": float(d_opt * 1e3),
                                          "mpjpe_hip_mm": float(mp_hip * 1e3), "mpjpe_ref_mm": float(mp_ref * 1e3),
                                          "evals_hip": ev.tolist(), "evals_ref": ref_ev.tolist()})
    assert d_mid < 0.5e-3 and d_opt < 0.5e-3, (d_mid, d_opt)
    assert abs(mp_hip - mp_ref) < 0.5e-3                 # the headline gate; the CPU oracle itself is within 0.05 mm here
    assert abs(mp_hip - mp_ref) < 0.1e-3, (mp_hip, mp_ref)
    eng.close()

    # the drop-in main(): same pickle, weights at the reference's hard-coded relative paths replaced by state dicts
    d = tmp_path / "chunk0"
    d.mkdir()
    with open(d / "test_data.pkl", "wb") as f:
        pickle.dump({k: list(data[k]) for k in ("estimated_local_skeleton", "gt_global_skeleton", "camera_pose_list", "heatmap_list")}, f)
    for device_metrics in (False, True):
        res = gopt.main(str(d), DEFAULT_CALIBRATION, final_smooth=True, global_vae_path=sd_g, local_vae_path=sd_l,
                        eps=torch.as_tensor(g["eps"]), device_metrics=device_metrics, **CLI)
        errors, est_seq, mid_local, opt_seq, gt_seq = res
        assert opt_seq.shape == (98, 15, 3)
        np.testing.assert_allclose(np.asarray(est_seq), g["est_smooth"], rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(np.asarray(gt_seq), g["gt_smooth"], rtol=1e-9, atol=1e-12)
        assert np.linalg.norm(opt_seq - g["opt_smooth"], axis=-1).mean() < 0.5e-3
        for k in errors:
            ref = g["err_smooth/" + k]
            if k.startswith("original") or k in ("aligned_original_mpjpe", "bone_length_aligned_original_mpjpe"):
                np.testing.assert_allclose(errors[k], ref, rtol=1e-8, atol=1e-11, err_msg=k)      # no optimisation involved
            else:
                assert np.abs(np.asarray(errors[k]) - ref).max() < 0.5e-3, (k, errors[k], ref)


# ------------------------------------------------------------------------------------------------------------------
# BASELINE configs[2..4]: size-independent properties + oracle spot checks at the full workload sizes
# ------------------------------------------------------------------------------------------------------------------
def _device_problem(eng, n_frames, starts, seed, n_dup, chunk=100):
    """Synthetic sequence resident in HBM + window table; the last n_dup windows repeat the first n_dup (same frames, same
    noise) so that identical windows in different tiles / workgroup waves can be compared bitwise."""
    import torch
    seq = synth.make_sequence_device(n_frames, seed=seed, device=eng.device, cam_jitter=(0.3, 0.002))
    starts = np.asarray(starts, dtype=np.int32).copy()
    B = len(starts)
    if n_dup:
        starts[B - n_dup:] = starts[:n_dup]
    g = torch.Generator().manual_seed(seed)
    eps = torch.randn(B, 2, 2048, generator=g)
    if n_dup:
        eps[B - n_dup:] = eps[:n_dup]
    n_chunks = (n_frames + chunk - 1) // chunk
    mb_c = torch.stack([eng.mean_bone_length(seq["est_local"][c * chunk:(c + 1) * chunk]) for c in range(n_chunks)])
    mb = mb_c[torch.as_tensor(starts // chunk, dtype=torch.long, device=eng.device)].contiguous()
    return {"seq": seq, "starts": starts, "f0": torch.as_tensor(starts, device=eng.device), "mb": mb,
            "eps_l": eps[:, 0].contiguous().to(eng.device), "eps_g": eps[:, 1].contiguous().to(eng.device), "n_dup": n_dup}


def _oracle_windows(p, sd_l, sd_g, windows, T=10, threads=8):
    """Both stages of the fp32 CPU oracle (optimizer.py:242-276 chained as optimizer.py:370-423 does) for the listed windows of
    problem `p`: {window: (global pose [T,15,3] f64, stats local stage, stats global stage)}.  Windows are independent: a thread pool
    (numpy releases the GIL in its matrix products) brings 48 windows to about half a minute on the box's host cores."""
    from concurrent.futures import ThreadPoolExecutor
    vae_l, vae_g, cam = O.fold_vae(sd_l), O.fold_vae(sd_g), oracle_camera()
    est_np, cams_np = p["seq"]["est_local_np"], p["seq"]["cams_np"]
    mb_np, eps_l, eps_g = p["mb"].cpu().numpy(), p["eps_l"].cpu().numpy(), p["eps_g"].cpu().numpy()
    heats = {int(b): p["seq"]["heat"][int(p["starts"][b]):int(p["starts"][b]) + T].cpu().numpy() for b in windows}

    def one(b):
        s = int(p["starts"][b])
        a, sa = O.optimize_stage(vae_l, cam, O.Weights(*W_LOCAL), est_np[s:s + T], heats[b], mb_np[b], eps_l[b])
        relo = O.relative_global(a, cams_np[s:s + T])
        c, sb = O.optimize_stage(vae_g, cam, O.Weights(*W_GLOBAL), relo.astype(np.float32), heats[b], mb_np[b], eps_g[b])
        return b, (O.to_global(c, cams_np[s:s + T]), sa, sb)
    with ThreadPoolExecutor(threads) as ex:
        return dict(ex.map(one, [int(b) for b in windows]))


def _chunk_sequences_against_oracle(glob_np, refs, p, chunks, per=12):
    """For every listed chunk: MPJPE (vs ground truth) of the merged + smoothed sequence (optimizer.py:425-450) built from the
    oracle's windows and from `glob_np`'s, and the mean distance between the two sequences -- where north_star's 0.5 mm applies."""
    out = []
    for c in chunks:
        w = list(range(c * per, (c + 1) * per))
        gt = p["seq"]["gt_global"][c * 100:c * 100 + 98]
        seq_or = final_smooth(merge_batches(np.stack([refs[b] for b in w])))
        seq_hip = final_smooth(merge_batches(glob_np[w[0]:w[-1] + 1]))
        out.append({"chunk": int(c), "mpjpe_oracle_mm": float(np.linalg.norm(seq_or - gt, axis=-1).mean() * 1e3),
                    "mpjpe_hip_mm": float(np.linalg.norm(seq_hip - gt, axis=-1).mean() * 1e3),
                    "sequence_diff_mm": float(np.linalg.norm(seq_or - seq_hip, axis=-1).mean() * 1e3)})
    return out


def _run(eng, p):
    return eng.optimize_windows(p["seq"]["est_local"], p["seq"]["cams"], p["seq"]["heat"], p["f0"], p["mb"], p["eps_l"], p["eps_g"],
                                _ew(W_LOCAL), _ew(W_GLOBAL))


def _check_properties(torch, eng, p, sd_l, sd_g, spot, tag, pose_tol_mm, loss_rtol):
    """status, evaluation bounds, Armijo on both stages, determinism, bitwise duplicates; then `spot` windows against the
    fp32 CPU oracle (both stages chained).  Returns (mid, glob, stats_numpy, {window: oracle global pose}).

    bf16 runs use the bf16-friendly gauge of the structured VAE (fixture friendly_vaes) and are held to the fp32 ORACLE: every
    spot window within pose_tol_mm and loss_rtol in final energy.  The per-window figure has a FLOOR set by the format, not by
    the optimiser: this VAE carries every pose coordinate on ONE channel with gain 1 as 1 + u (u = metres / 0.3); each of the
    five bf16 activations between the decoder layers rounds it to 8 significant bits, i.e. to 1.2 / 2.3 mm steps of pose (values
    in [0.5, 1) / [1, 2)): 0.54 mm rms per rounding and coordinate, 1.2 mm after five, ~1.9 mm mean 3-D joint distance before
    anything else happens.  Measured: 2.0-3.2 mm on every window (uniform: noise, not diverged trajectories); the sequence
    MPJPE -- where that zero-mean noise averages out -- is where north_star's 0.5 mm is asserted (measured 0.01-0.06 mm).
    A fitted VAE spreads each coordinate over many channels: there bf16 and fp32 agree to 0.001 mm MPJPE (bench.py)."""
    from globalegomocap_amd.engine import stats_to_numpy, LOCAL_STAGE, GLOBAL_STAGE
    B, n_dup = len(p["starts"]), p["n_dup"]
    mid, glob, stats = _run(eng, p)
    sn = stats_to_numpy(stats)
    assert (sn["status"] == 1).all(), tag
    assert (sn["func_evals"] <= 32).all() and (sn["func_evals"] >= 1).all() and (sn["n_iter"] <= 25).all(), tag
    assert np.isfinite(glob.cpu().numpy()).all() and np.isfinite(sn["final_loss"]).all(), tag
    # the optimiser really works at this size (not a batch of first-test exits)
    assert sn["func_evals"][:B].mean() > 25 and sn["func_evals"][B:].mean() > 8, (tag, sn["func_evals"][:B].mean(), sn["func_evals"][B:].mean())
    # determinism and independence of the windows: bitwise
    mid2, glob2, stats2 = _run(eng, p)
    assert torch.equal(mid, mid2) and torch.equal(glob, glob2) and torch.equal(stats, stats2), tag
    if n_dup:
        assert torch.equal(mid[:n_dup], mid[B - n_dup:]) and torch.equal(glob[:n_dup], glob[B - n_dup:]), tag
        assert torch.equal(stats[:n_dup], stats[B - n_dup:B]) and torch.equal(stats[B:B + n_dup], stats[2 * B - n_dup:]), tag
    # Armijo: the accepted energy never exceeds the energy at the stage's starting point (checked on a sample of windows)
    idx = np.unique(np.concatenate([np.arange(0, B, max(1, B // 64)), [B - 1]]))
    ii = torch.as_tensor(idx, device=eng.device)
    T = 10
    fr = (p["f0"][ii].long()[:, None] + torch.arange(T, device=eng.device)[None])
    pose_l = p["seq"]["est_local"][fr]                                       # [n,T,15,3]
    small = _engine(len(idx), sd_l, sd_g, "f32")                                 # reference energies in fp32
    _, _, z0 = small.encode(LOCAL_STAGE, pose_l.reshape(len(idx), T, 45), p["eps_l"][ii])
    E0, _, _, _ = small.energy_grad(LOCAL_STAGE, z0, pose_l, p["mb"][ii], _ew(W_LOCAL), p["seq"]["heat"], p["f0"][ii])
    slack = 1e-6 if eng.precision == "f32" else 2e-3                        # bf16 products move the energies themselves a little
    assert (sn["final_loss"][idx] <= E0.cpu().numpy().astype(np.float32) + slack * (1 + np.abs(E0.cpu().numpy()))).all(), tag
    cw = p["seq"]["cams"][fr]
    M = torch.linalg.inv(cw[:, :1]) @ cw                                         # C0^-1 C_t
    rel = (torch.einsum("btij,btkj->btki", M[..., :3, :3], mid[ii].double()) + M[..., None, :3, 3]).float()
    _, _, z0g = small.encode(GLOBAL_STAGE, rel.reshape(len(idx), T, 45), p["eps_g"][ii])
    E0g, _, _, _ = small.energy_grad(GLOBAL_STAGE, z0g, rel, p["mb"][ii], _ew(W_GLOBAL))
    assert (sn["final_loss"][B + idx] <= E0g.cpu().numpy().astype(np.float32) + slack * (1 + np.abs(E0g.cpu().numpy()))).all(), tag
    small.close()
    # oracle spot check: both stages chained on the CPU, fp32
    T = 10
    glob_np = glob.cpu().numpy()
    rep = []
    refs = {}
    for b, (ref, sa, sb) in _oracle_windows(p, sd_l, sd_g, spot).items():
        refs[int(b)] = ref
        d = np.linalg.norm(glob_np[b] - ref, axis=-1).mean()
        rep.append({"window": int(b), "diff_mm": float(d * 1e3), "evals": [int(sn["func_evals"][b]), int(sn["func_evals"][B + b])],
                    "oracle_evals": [int(sa["func_evals"]), int(sb["func_evals"])],
                    "loss": [float(sn["final_loss"][b]), float(sn["final_loss"][B + b])], "oracle_loss": [float(sa["loss"]), float(sb["loss"])]})
        assert d * 1e3 < pose_tol_mm, (tag, rep[-1])
        assert abs(sn["final_loss"][b] - sa["loss"]) <= loss_rtol * abs(sa["loss"]), (tag, rep[-1])
        if eng.precision == "f32":
            assert abs(int(sn["func_evals"][b]) - sa["func_evals"]) <= 3, (tag, rep[-1])
    print(tag, "oracle spot check:", ["%d: %.3f mm" % (r["window"], r["diff_mm"]) for r in rep])
    _report("spot_%s.json" % tag, rep)
    return mid, glob, sn, refs


def _seq_mpjpe(glob, p, n_chunks, per):
    g = glob.cpu().numpy()
    out, gt = [], []
    for c in range(n_chunks):
        m = final_smooth(merge_batches(g[c * per:(c + 1) * per]))
        out.append(m)
        gt.append(p["seq"]["gt_global"][c * 100:c * 100 + m.shape[0]])
    return float(np.linalg.norm(np.concatenate(out) - np.concatenate(gt), axis=-1).mean())


def test_config2_all_sequences_in_one_call_bf16(torch_cuda, friendly_vaes, tmp_path):
    """BASELINE configs[2]: "all 5 test-sequence shapes concurrently on 1 MI355X, bf16 VAE decoder / fp32 energy" -- five
    sequences of 20 + 27 + 27 + 27 + 27 chunks = 1536 windows, every window in ONE device call.
    bf16 against the fp32 CPU ORACLE (north_star's tolerance): the 48 windows of FOUR chunks spread over the batch (0, 42, 85, 125)
    each within the bf16 activation noise floor (see _check_properties), and the MPJPE of each of those chunks' merged + smoothed
    sequences within 0.5 mm of the oracle's; over all 126 chunks bf16 vs fp32 HIP (the path pinned to the reference at 0.1 mm by the
    golden tests) within 0.5 mm as well."""
    torch = torch_cuda
    sd_l, sd_g = friendly_vaes
    n_chunks, per = 128, 12
    B = n_chunks * per
    starts = np.concatenate([c * 100 + window_starts(100) for c in range(n_chunks)])
    eng = _engine(B, sd_l, sd_g, "f32")
    p = _device_problem(eng, n_chunks * 100, starts, seed=202, n_dup=24)
    _, glob_f32, _ = _run(eng, p)
    mp_f32 = _seq_mpjpe(glob_f32, p, n_chunks - 2, per)          # (the last two chunks hold the duplicated windows)
    eng.set_precision("bf16")
    # FOUR chunks spread over the batch (the last two chunks hold the duplicated windows) go to the oracle: 48 windows
    or_chunks = (0, 42, 85, 125)
    spot = tuple(b for c in or_chunks for b in range(c * per, (c + 1) * per)) + (640,)
    mid, glob, sn, refs = _check_properties(torch, eng, p, sd_l, sd_g, spot=spot, tag="configs2_bf16", pose_tol_mm=4.5, loss_rtol=2e-2)
    mp_bf16 = _seq_mpjpe(glob, p, n_chunks - 2, per)
    # per chunk: the oracle's merged + smoothed sequence against the bf16 HIP one, both against the ground truth
    seqs = _chunk_sequences_against_oracle(glob.cpu().numpy(), refs, p, or_chunks, per)
    print("configs[2]: MPJPE f32 %.3f mm, bf16 %.3f mm over %d frames; chunks vs oracle: %s"
          % (mp_f32 * 1e3, mp_bf16 * 1e3, (n_chunks - 2) * 98, seqs))
    _report("configs2_mpjpe.json", {"mpjpe_f32_mm": mp_f32 * 1e3, "mpjpe_bf16_mm": mp_bf16 * 1e3, "chunks_vs_oracle": seqs})
    for r in seqs:                                                # north_star: MPJPE within 0.5 mm of the reference path
        # (frame by frame the two sequences differ by this VAE's bf16 noise floor -- 2-3 mm per window, see _check_properties, 1.7 mm
        # after the merge and the smoothing; it is zero-mean, which is what the MPJPE comparison shows)
        assert abs(r["mpjpe_hip_mm"] - r["mpjpe_oracle_mm"]) < 0.5 and r["sequence_diff_mm"] < 3.0, r
    assert abs(mp_bf16 - mp_f32) < 0.5e-3, (mp_bf16, mp_f32)
    eng.close()

    # and through the reference-shaped entry point: five sequence directories -> optimize_sequences, one device call
    from globalegomocap_amd import whole_sequence as WS
    from globalegomocap_amd.optimizer import SequenceOptimizer
    pool = []
    for c in range(27):                                          # 27 distinct chunk pickles, linked into the five sequences
        d = tmp_path / "pool" / ("chunk_%d" % c)
        d.mkdir(parents=True)
        sq = synth.make_sequence(100, seed=900 + c, cam_jitter=(0.3, 0.002))
        with open(d / "test_data.pkl", "wb") as f:
            pickle.dump({k: sq[k] for k in ("estimated_local_skeleton", "gt_global_skeleton", "camera_pose_list", "heatmap_list")}, f)
        pool.append(d)
    dirs = []
    for si, n in enumerate((20, 27, 27, 27, 27)):
        sd_dir = tmp_path / ("seq%d" % si)
        sd_dir.mkdir()
        for c in range(n):
            os.symlink(pool[(c + 3 * si) % 27], sd_dir / ("chunk_%d" % c))
        dirs.append(str(sd_dir))
    opt = SequenceOptimizer(DEFAULT_CALIBRATION, sd_g, sd_l, max_windows=B)
    calls = []
    real = opt.engine.optimize_windows

    def counting(*a, **k):
        calls.append(int(a[3].shape[0]))
        return real(*a, **k)
    opt.engine.optimize_windows = counting
    res = {}
    for mode in ("f32", "bf16"):
        opt.engine.set_precision(mode)
        torch.manual_seed(77)
        res[mode] = WS.optimize_sequences(dirs, DEFAULT_CALIBRATION, optimizer=opt, verbose=False)
    assert calls == [B, B], calls                                # ONE call with all 1536 windows per run
    for (sm_f, per_f, _, opt_f, gt_f), (sm_b, per_b, _, opt_b, _) in zip(res["f32"], res["bf16"]):
        assert len(per_b) == len(per_f) and np.isfinite(list(v for k, v in sm_b.items() if k != "joints_error")).all()
        assert sm_b["optimized_global_mpjpe"] < sm_b["original_global_mpjpe"] - 5e-3          # the optimisation helps (metres)
        assert abs(sm_b["optimized_global_mpjpe"] - sm_f["optimized_global_mpjpe"]) < 0.5e-3
        assert np.asarray(opt_b).shape == np.asarray(gt_f).shape
    opt.engine.close()


def test_bf16_on_fitted_vae_against_the_oracle(torch_cuda):
    """bf16 decoder mode on REALISTIC weights against the fp32 CPU oracle (optimizer.py:242-276 both stages chained, restated in
    oracle/np_oracle.py).  The structured VAEs of the other configs tests carry every pose coordinate on one channel, which bf16
    resolves to ~1-2 mm per rounding (the 3.5 mm per-window gate there); a FITTED network spreads a coordinate over many channels.
    Two full-size VAEs are fitted on the device (`fit_vae_device`, the bench's recipe and seeds: ~3 s each), BASELINE configs[2]'s
    1536 windows run in bf16 AND in fp32 (the control: two fp32 implementations of these 30-evaluation L-BFGS runs already part by a
    few tenths of a mm per window), and the 48 windows of FOUR chunks spread over the batch (0, 42, 85, 127) are compared with the
    oracle, window by window and as each chunk's merged + smoothed sequence.

    Measured (round 4, profiles/parity_r04_bf16_fitted_vs_oracle.json): bf16 0.5-2.4 mm per window (mean 1.3 mm) -- NOT the 1.0 mm
    the round-3 review hoped for: a decoded coordinate of ~1 m is a sum of ~200 bf16 products, each good to 2^-9, so ~1 mm of
    zero-mean noise per coordinate is what the format carries whatever the weights are.  It IS zero-mean: the chunk's merged +
    smoothed sequence -- where north_star's 0.5 mm applies -- differs from the oracle's by 0.06 mm MPJPE.
    Asserted: bf16 per window <= 1.5 mm on average (<= 5 mm each), fp32 median <= 0.3 mm (<= 5 mm each: a trajectory that parts at a texel
    edge), |dMPJPE| of every one of the four chunks' sequences <= 0.5 mm for both."""
    torch = torch_cuda
    from globalegomocap_amd.vae_train import fit_vae_device
    from globalegomocap_amd.engine import stats_to_numpy
    sds = []
    for seed, relative in ((101, False), (102, True)):
        win = synth.make_training_windows(4096, FULL.seq_len, seed)
        if relative:                                   # relative-global poses drift with the camera: 4 mm / frame along x (bench.py)
            win = win.reshape(-1, FULL.seq_len, 15, 3).copy()
            win[..., 0] += (0.004 * np.arange(FULL.seq_len))[None, :, None]
            win = win.reshape(-1, FULL.seq_len, 45)
        sd, err = fit_vae_device(FULL, win, steps=2000, batch=128, lr=2e-3, kl_weight=0.01, seed=seed, latent_gain=8.0)
        assert err < 6e-3, ("the device fit did not converge", seed, err)
        sds.append({k: np.asarray(v) for k, v in sd.items()})
    sd_l, sd_g = sds
    n_chunks, per, T = 128, 12, 10
    B = n_chunks * per
    starts = np.concatenate([c * 100 + window_starts(100) for c in range(n_chunks)])
    eng = _engine(B, sd_l, sd_g, "bf16")
    p = _device_problem(eng, n_chunks * 100, starts, seed=505, n_dup=0)
    glob_np = {}
    for mode in ("bf16", "f32"):
        eng.set_precision(mode)
        mid, glob, stats = _run(eng, p)
        sn = stats_to_numpy(stats)
        assert (sn["status"] == 1).all() and np.isfinite(glob.cpu().numpy()).all(), mode
        glob_np[mode] = glob.cpu().numpy()
    # FOUR chunks spread over the batch against the oracle (48 windows): per window, and per chunk as the merged + smoothed sequence
    or_chunks = (0, 42, 85, 127)
    spot = tuple(b for c in or_chunks for b in range(c * per, (c + 1) * per))
    refs, rep = {}, []
    for b, (ref, sa, sb) in _oracle_windows(p, sd_l, sd_g, spot).items():
        refs[b] = ref
        rep.append({"window": b, "bf16_diff_mm": float(np.linalg.norm(glob_np["bf16"][b] - ref, axis=-1).mean() * 1e3),
                    "f32_diff_mm": float(np.linalg.norm(glob_np["f32"][b] - ref, axis=-1).mean() * 1e3),
                    "oracle_evals": [int(sa["func_evals"]), int(sb["func_evals"])]})
    seqs = {m: _chunk_sequences_against_oracle(g, refs, p, or_chunks, per) for m, g in glob_np.items()}
    d16 = np.array([r["bf16_diff_mm"] for r in rep])
    d32 = np.array([r["f32_diff_mm"] for r in rep])
    _report("bf16_fitted_vs_oracle.json", {"windows": rep, "chunks_vs_oracle": seqs,
                                           "bf16_per_window_mm": {"min": d16.min(), "mean": d16.mean(), "max": d16.max()},
                                           "f32_per_window_mm": {"min": d32.min(), "mean": d32.mean(), "max": d32.max()}})
    print("fitted VAEs vs oracle, per window: bf16 %.2f / %.2f / %.2f mm (min / mean / max), fp32 %.2f / %.2f / %.2f mm; chunks: %s"
          % (d16.min(), d16.mean(), d16.max(), d32.min(), d32.mean(), d32.max(), seqs))
    # (per-window maxima are the fragile statistic here: one L-BFGS trajectory that crosses a heat-map texel edge on the other side
    # ends millimetres away at nearly the same energy -- in fp32 as well; seen: 4.3 mm on one of 24 fp32 windows)
    assert d16.max() <= 5.0 and d16.mean() <= 1.5, rep
    assert d32.max() <= 5.0 and np.median(d32) <= 0.3 and d32.mean() <= 0.8, rep
    for m in ("bf16", "f32"):
        for r in seqs[m]:
            assert abs(r["mpjpe_hip_mm"] - r["mpjpe_oracle_mm"]) <= 0.5, (m, r)
    eng.close()


def test_config3_shard_8192_windows_bf16(torch_cuda, friendly_vaes):
    """BASELINE configs[3]: 64k synthetic windows over 8 GPUs, bf16 -- the per-GPU shard: 8192 windows in one call; six windows
    spread over the batch against the fp32 CPU oracle."""
    torch = torch_cuda
    sd_l, sd_g = friendly_vaes
    B, n_frames = 8192, 12000
    rng = np.random.default_rng(303)
    starts = rng.integers(0, n_frames - 10, B)
    eng = _engine(B, sd_l, sd_g, "bf16")
    p = _device_problem(eng, n_frames, starts, seed=303, n_dup=128)
    _check_properties(torch, eng, p, sd_l, sd_g, spot=(0, 127, 128, 4095, 4096, 8063), tag="configs3_bf16", pose_tol_mm=4.5,
                      loss_rtol=2e-2)
    eng.close()


@pytest.mark.parametrize("precision,pose_tol_mm,loss_rtol", [("f32", 2.0, 2e-3), ("bf16", 4.5, 2e-2)])
def test_config4_shard_of_a_continuous_stream(torch_cuda, friendly_vaes, precision, pose_tol_mm, loss_rtol):
    """BASELINE configs[4]: a 100k-frame stream over 8 GPUs -- the per-GPU shard: 1563 overlapping windows (stride 8) of ONE
    continuous 12 506-frame sequence, frames stored once, no chunk structure."""
    torch = torch_cuda
    sd_l, sd_g = friendly_vaes
    B = 1563
    starts = 8 * np.arange(B)
    n_frames = int(starts[-1]) + 10
    eng = _engine(B, sd_l, sd_g, precision)
    p = _device_problem(eng, n_frames, starts, seed=404, n_dup=0)
    mid, glob, sn, _ = _check_properties(torch, eng, p, sd_l, sd_g, spot=(0, 1, 700, 701, 1561, 1562), tag="configs4_" + precision,
                                         pose_tol_mm=pose_tol_mm, loss_rtol=loss_rtol)
    # one continuous sequence: merge ALL windows as one chunk (overlap 2) and smooth -> [8*B+2] frames
    merged = final_smooth(merge_batches(glob.cpu().numpy()))
    assert merged.shape == (8 * B + 2, 15, 3)
    gt = p["seq"]["gt_global"][:merged.shape[0]]
    homo = np.concatenate([p["seq"]["est_local_np"][:merged.shape[0]], np.ones((merged.shape[0], 15, 1))], -1)
    est = np.einsum("nij,nkj->nki", p["seq"]["cams_np"][:merged.shape[0]], homo)[..., :3]
    mp_opt = np.linalg.norm(merged - gt, axis=-1).mean()
    mp_in = np.linalg.norm(est - gt, axis=-1).mean()
    print("configs[4] %s: MPJPE %.2f -> %.2f mm over %d frames" % (precision, mp_in * 1e3, mp_opt * 1e3, merged.shape[0]))
    assert mp_opt < mp_in - 5e-3
    eng.close()


@pytest.mark.parametrize("B", [4360, 8192])
def test_two_lanes_are_bitwise_one_lane(torch_cuda, friendly_vaes, B):
    """Large bf16 batches run as two half-batches on two streams, half an evaluation round apart (gem_set_lanes / windows_dual
    in csrc/gem_api.hip), so that one half's HBM-bound L-BFGS advance overlaps the other half's matrix-bound kernels.  Windows do
    not interact and from 4352 windows on no product is cut along K in the full batch or in its halves: mid-local poses, global
    poses, statistics and closure traces must be bit for bit those of ONE lane -- eagerly and replayed from a hipGraph."""
    torch = torch_cuda
    sd_l, sd_g = friendly_vaes
    rng = np.random.default_rng(B)
    n_frames = 6000
    starts = rng.integers(0, n_frames - 10, B)
    one = _engine(B, sd_l, sd_g, "bf16")
    one.set_lanes(0)
    p = _device_problem(one, n_frames, starts, seed=B, n_dup=0)
    ref = [t.clone() for t in _run(one, p)]
    tr_ref = one.read_trace(B, 8)
    one.close()
    two = _engine(B, sd_l, sd_g, "bf16")
    two.set_lanes(4352)                                  # two lanes from 4352 windows on (one lane is the default since round 4)
    for graphs in (False, True):
        two.enable_graphs(graphs)
        for k in range(3 if graphs else 1):
            got = _run(two, p)
            torch.cuda.synchronize()
            for a_, b_ in zip(got, ref):
                assert torch.equal(a_, b_), (graphs, k)
        assert np.array_equal(two.read_trace(B, 8), tr_ref, equal_nan=True)
    assert two.graph_stats()["replays"] >= 2
    two.close()


def test_device_wide_barrier_experiment_is_bitwise_the_product(torch_cuda, full_vaes, monkeypatch):
    """The experiment behind DESIGN.md section 4's "a device-wide barrier against a launch boundary" (GEM_DEV=1
    GEM_FUSE_BWD_LBFGS=1: the backward front product and lbfgs_advance of every fp32 round as ONE launch with a grid barrier in
    between, csrc/lbfgs.hip rows_bwd_lbfgs_kernel): it is only a timing argument if it computes the same thing -- 240 windows, both
    stages, bitwise the product path's poses and statistics."""
    torch = torch_cuda
    data, sd_l, sd_g, w_l, w_g = full_vaes
    n_chunks = 20
    starts = np.concatenate([c * 100 + window_starts(100) for c in range(n_chunks)])
    eng = _engine(len(starts), sd_l, sd_g, "f32")
    p = _device_problem(eng, n_chunks * 100, starts, seed=707, n_dup=0)
    ref = [t.clone() for t in _run(eng, p)]
    monkeypatch.setenv("GEM_DEV", "1")
    monkeypatch.setenv("GEM_FUSE_BWD_LBFGS", "1")
    out = [t.clone() for t in _run(eng, p)]
    torch.cuda.synchronize()
    monkeypatch.delenv("GEM_FUSE_BWD_LBFGS")
    assert all(torch.equal(a, b) for a, b in zip(ref, out))
    from globalegomocap_amd.engine import stats_to_numpy
    assert stats_to_numpy(out[2])["finished"].all()
    eng.close()


# ------------------------------------------------------------------------------------------------------------------
# hipGraph replay (BASELINE configs[4])
# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("precision", ["f32", "bf16"])
def test_graph_replay_is_bitwise_the_eager_path(torch_cuda, full_vaes, precision):
    """A whole optimize_windows call replayed from a hipGraph (gem_graph_enable): bitwise the eager result on the 240-window
    batch of BASELINE configs[1], replays really happen, a changed input pointer falls back to a new capture, and two engines
    replaying on two streams do not disturb each other."""
    import time
    torch = torch_cuda
    data, sd_l, sd_g, w_l, w_g = full_vaes
    n_chunks, per = 20, 12
    B = n_chunks * per
    starts = np.concatenate([c * 100 + window_starts(100) for c in range(n_chunks)])
    eng = _engine(B, sd_l, sd_g, precision)
    p = _device_problem(eng, n_chunks * 100, starts, seed=505, n_dup=0)
    mid0, glob0, st0 = (t.clone() for t in _run(eng, p))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        _run(eng, p)
    t_eager_enqueue = (time.perf_counter() - t0) / 3
    torch.cuda.synchronize()
    eng.enable_graphs(True)
    for k in range(4):                              # call 1 eager (warm-up), call 2 captures, calls 3-4 replay
        mid, glob, st = _run(eng, p)
        torch.cuda.synchronize()
        assert torch.equal(mid, mid0) and torch.equal(glob, glob0) and torch.equal(st, st0), (precision, k)
    gs = eng.graph_stats()
    assert gs["captures"] == 1 and gs["replays"] == 3, gs
    t0 = time.perf_counter()
    for _ in range(3):
        _run(eng, p)
    t_graph_enqueue = (time.perf_counter() - t0) / 3
    torch.cuda.synchronize()
    print("%s: host enqueue per 240-window call: eager %.2f ms, graph replay %.3f ms" % (precision, t_eager_enqueue * 1e3, t_graph_enqueue * 1e3))
    _report("graph_enqueue_%s.json" % precision, {"eager_ms": t_eager_enqueue * 1e3, "graph_ms": t_graph_enqueue * 1e3})
    # (a printed / recorded metric, not an assertion: host wall-clock on a shared box; round 2 measured 0.10-0.22 ms against
    # 1.9 ms eager.  What IS asserted is that the calls were replays.)
    # another eps tensor = another signature: eager once, then captured again; results follow the new input
    p2 = dict(p)
    p2["eps_l"] = p["eps_l"].clone()
    p2["eps_l"][0] += 0.5
    r_a = [t.clone() for t in _run(eng, p2)]
    r_b = [t.clone() for t in _run(eng, p2)]
    torch.cuda.synchronize()
    assert eng.graph_stats()["captures"] == 2
    assert all(torch.equal(x, y) for x, y in zip(r_a, r_b)) and not torch.equal(r_a[1][0], glob0[0]) and torch.equal(r_a[1][1:], glob0[1:])
    # two engines, two graph streams, interleaved replays
    eng2 = _engine(B, sd_l, sd_g, precision)
    eng2.enable_graphs(True)
    p3 = _device_problem(eng2, n_chunks * 100, starts, seed=606, n_dup=0)
    ref3 = [t.clone() for t in _run(eng2, p3)]
    _run(eng2, p3)
    torch.cuda.synchronize()
    for _ in range(3):
        a = _run(eng, p)
        b = _run(eng2, p3)
    torch.cuda.synchronize()
    assert torch.equal(a[1], glob0) and torch.equal(b[1], ref3[1]) and torch.equal(b[2], ref3[2])
    eng.close(); eng2.close()
