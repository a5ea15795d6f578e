"""GPU parity tests of the sequence post-processing calls: overlap merge + Gaussian
smoothing and the 18-entry error report, against the host mirrors (which tests/test_host_cpu.py pins to
the reference's golden run) and against the golden run itself."""
import os

import numpy as np
import pytest

from globalegomocap_amd import sequence, synth
from globalegomocap_amd.camera import FisheyeCamera, DEFAULT_CALIBRATION
from globalegomocap_amd.errors import calculate_errors
from globalegomocap_amd.skeleton import MEAN3D_MM
from helpers import TINY

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def engine():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("gpu tests need a HIP device")
    from globalegomocap_amd.engine import WindowEngine
    return WindowEngine(TINY, FisheyeCamera.from_json(DEFAULT_CALIBRATION), max_windows=8)


def _sequences(F, seed, noise=(0.03, 0.02, 0.01)):
    """gt = walking mean skeleton with per-frame articulation; est/mid/opt = rotated, scaled, noisy copies."""
    rng = np.random.default_rng(seed)
    base = MEAN3D_MM.T / 1000.0
    t = np.arange(F)[:, None, None]
    gt = base[None] + 0.05 * np.sin(0.1 * t + rng.uniform(0, 6, (1, 15, 3))) + np.array([0.01, 0.0, 0.002]) * t
    out = []
    for k, s in enumerate(noise):
        a = 0.2 * (k + 1)
        Rz = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1.0]])
        out.append((1.0 + 0.05 * k) * gt @ Rz + rng.normal(0, s, gt.shape) + rng.normal(0, 0.1, (1, 1, 3)))
    return out[0], out[1], out[2], gt


@pytest.mark.parametrize("n_chunks,wpc,overlap,smooth", [(3, 12, 2, True), (1, 1, 2, True), (2, 5, 2, False), (2, 4, 0, True),
                                                         (1, 3, 5, False)])
def test_merge_and_smooth_against_host_mirror(engine, n_chunks, wpc, overlap, smooth):
    rng = np.random.default_rng(7)
    w = rng.normal(size=(n_chunks * wpc, 10, 15, 3))
    got = engine.merge_windows(w, n_chunks, overlap=overlap, smooth=smooth).cpu().numpy()
    ref = []
    for c in range(n_chunks):
        m = sequence.merge_batches(w[c * wpc:(c + 1) * wpc], overlap=overlap)
        ref.append(sequence.final_smooth(m) if smooth else m)
    ref = np.concatenate(ref)
    assert got.shape == ref.shape
    np.testing.assert_allclose(got, ref, rtol=0, atol=1e-14)      # float64 both sides; only the summation order differs


def test_merge_rejects_bad_arguments(engine):
    from globalegomocap_amd._capi import GemError
    with pytest.raises(ValueError):
        engine.merge_windows(np.zeros((5, 10, 15, 3)), 2)
    with pytest.raises(GemError):
        engine.merge_windows(np.zeros((4, 10, 15, 3)), 2, overlap=6)


@pytest.mark.parametrize("F,seed", [(98, 1), (257, 2), (1, 3), (2000, 4)])
def test_error_report_against_host_mirror(engine, F, seed):
    est, mid, opt, gt = _sequences(F, seed)
    ref = calculate_errors(est, mid, opt, gt)
    got = engine.calculate_errors(est, mid, opt, gt)
    assert list(got.keys()) == list(ref.keys())
    for k in ref:
        np.testing.assert_allclose(got[k], ref[k], rtol=1e-9, atol=1e-12, err_msg=k)


def test_error_report_is_reproducible_and_handles_reflections(engine):
    est, mid, opt, gt = _sequences(130, 9)
    est = est * np.array([1.0, 1.0, -1.0])          # mirrored estimate: det(V)det(W) < 0 -> the reflection fix is exercised
    a = engine.calculate_errors_device(est, mid, opt, gt).cpu().numpy()
    b = engine.calculate_errors_device(est, mid, opt, gt).cpu().numpy()
    assert np.array_equal(a, b)
    ref = calculate_errors(est, mid, opt, gt)
    got = engine.calculate_errors(est, mid, opt, gt)
    for k in ref:
        np.testing.assert_allclose(got[k], ref[k], rtol=1e-9, atol=1e-12, err_msg=k)


def test_error_report_against_reference_golden(engine, golden):
    g = golden("pipeline_tiny")
    data = synth.make_sequence(n_frames=100, seed=int(g["seq_seed"]), with_heatmaps=False)
    cams = np.asarray(data["camera_pose_list"])
    for tag in ("smooth", "raw"):
        mid = g["mid_local_" + tag] + cams[:98, :3, 3][:, None, :]      # see tests/test_host_cpu.py for why this is exact
        e = engine.calculate_errors(g["est_" + tag], mid, g["opt_" + tag], g["gt_" + tag])
        for k, v in e.items():
            ref = g["err_%s/%s" % (tag, k)]
            tol = dict(rtol=1e-5, atol=1e-7) if "mid" in k else dict(rtol=1e-9, atol=1e-12)
            np.testing.assert_allclose(v, ref, err_msg=k, **tol)


def test_error_report_rejects_bad_arguments(engine):
    est, mid, opt, gt = _sequences(10, 5)
    with pytest.raises(AssertionError):
        engine.calculate_errors(est[:5], mid, opt, gt)


@pytest.mark.parametrize("smooth", [True, False])
def test_main_with_device_metrics_matches_host_metrics(golden, tmp_path, smooth):
    """The main() mirror with merge / smoothing / calculate_errors on the device vs the same run with numpy."""
    import pickle
    import torch
    from globalegomocap_amd import optimizer as gopt
    from helpers import sd_from_npz
    g = golden("pipeline_tiny")
    lt = golden("lbfgs_tiny")
    data = synth.make_sequence(n_frames=100, seed=int(g["seq_seed"]))
    d = tmp_path / "chunk0"
    d.mkdir()
    with open(d / "test_data.pkl", "wb") as f:
        pickle.dump({k: data[k] for k in ("estimated_local_skeleton", "gt_global_skeleton", "camera_pose_list", "heatmap_list")}, f)
    torch.manual_seed(int(g["eps_seed"]))
    eps = torch.randn(24, 32)
    kw = dict(final_smooth=smooth, global_vae_path=sd_from_npz(lt, "global/"), local_vae_path=sd_from_npz(lt, "local/"), eps=eps)
    args = (str(d), DEFAULT_CALIBRATION, 0.0, 0.0, float(g["smooth"]), 0.01, float(g["weight_3d"]), 0.01)
    host = gopt.main(*args, **kw)
    dev = gopt.main(*args, device_metrics=True, **kw)
    np.testing.assert_allclose(dev[3], host[3], rtol=0, atol=1e-13)           # final_optimized_seq (the run itself is bitwise repeatable)
    assert list(dev[0].keys()) == list(host[0].keys())
    for k in host[0]:
        np.testing.assert_allclose(dev[0][k], host[0][k], rtol=1e-9, atol=1e-12, err_msg=k)
    tag = "smooth" if smooth else "raw"
    assert abs(dev[0]["optimized_global_mpjpe"] - float(g["err_%s/optimized_global_mpjpe" % tag])) < 0.5e-3


# ------------------------------------------------------------------------------------------------ input lifting (8f.2)
@pytest.mark.parametrize("tag", ["default", "alt"])
def test_input_lifting_against_reference_golden(golden, tag):
    from globalegomocap_amd.camera import ALT_CALIBRATION
    from globalegomocap_amd.engine import WindowEngine
    g = golden("lift")
    cam = FisheyeCamera.from_json(DEFAULT_CALIBRATION if tag == "default" else ALT_CALIBRATION)
    eng = WindowEngine(TINY, cam, max_windows=4)
    o64, o32 = eng.lift_skeleton(g["heat"].astype(np.float32), g["depth"])
    ref = g["skeleton_" + tag]
    np.testing.assert_allclose(o64.cpu().numpy(), ref, rtol=1e-12, atol=1e-14)
    assert np.array_equal(o32.cpu().numpy(), ref.astype(np.float32)) or \
        np.max(np.abs(o32.cpu().numpy() - ref.astype(np.float32))) < 1e-6


def test_input_lifting_against_oracle_random_and_nan(engine):
    from oracle import np_oracle as O
    rng = np.random.default_rng(5)
    F = 12
    heat = rng.normal(0.2, 1.0, (F, 64, 64, 15)).astype(np.float32)
    heat[0, 5, 6, 2] = np.nan               # numpy's argmax lets NaN win, the `max > 0` mask then zeroes the prediction
    heat[1] = np.round(heat[1])             # many ties
    depth = rng.uniform(0.2, 2.0, (F, 15))
    o64, o32 = engine.lift_skeleton(heat, depth)
    cam = engine.camera
    ref = np.stack([O.lift_skeleton(heat[f], depth[f], cam.poly_c2w, cam.cx, cam.cy) for f in range(F)])
    np.testing.assert_allclose(o64.cpu().numpy(), ref, rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(o32.cpu().numpy(), ref, rtol=1e-6, atol=1e-7)


def test_input_lifting_feeds_the_window_optimiser_shapes_and_errors(engine):
    from globalegomocap_amd._capi import GemError
    with pytest.raises(ValueError):
        engine.lift_skeleton(np.zeros((2, 32, 64, 15), np.float32), np.ones((2, 15)))
    o64, o32 = engine.lift_skeleton(np.zeros((0, 64, 64, 15), np.float32), np.ones((0, 15)))
    assert o64.shape == (0, 15, 3) and o32.shape == (0, 15, 3)
    heat = np.random.default_rng(0).uniform(0, 1, (3, 64, 64, 15)).astype(np.float32)
    _, a = engine.lift_skeleton(heat, np.ones((3, 15)), want_f64=False)
    assert a.dtype.is_floating_point and tuple(a.shape) == (3, 15, 3)
    assert np.allclose(np.linalg.norm(a.cpu().numpy(), axis=-1), 1.0, atol=1e-6)      # unit depth -> unit rays
    with pytest.raises(GemError):
        engine.lift_skeleton(heat, np.ones((3, 15)), upscale=0)


# ------------------------------------------------------------------------------------------------ whole-sequence driver
def test_whole_sequence_driver_matches_per_chunk_main(golden, tmp_path, capsys):
    """All chunk directories in one batched call vs the reference's loop of main() per chunk (same noise order)."""
    import pickle
    import torch
    from globalegomocap_amd import optimizer as gopt, whole_sequence as ws
    from helpers import sd_from_npz
    lt = golden("lbfgs_tiny")
    sd_g, sd_l = sd_from_npz(lt, "global/"), sd_from_npz(lt, "local/")
    for i, n in ((1, 100), (2, 100), (10, 60)):
        d = tmp_path / ("seq_%d" % i)
        d.mkdir()
        data = synth.make_sequence(n_frames=n, seed=40 + i)
        with open(d / "test_data.pkl", "wb") as f:
            pickle.dump(synth.reference_pickle_dict(data), f)          # as the reference's tool writes it
    kw = dict(global_vae_path=sd_g, local_vae_path=sd_l)
    torch.manual_seed(77)
    per_chunk = [gopt.main(p, DEFAULT_CALIBRATION, 0.0, 0.0, 0.001, 0.01, 0.01, 0.01, final_smooth=True, **kw)
                 for p in ws.list_chunks(str(tmp_path))]
    torch.manual_seed(77)
    summary, results, est, opt, gt = ws.optimize_directory(str(tmp_path), DEFAULT_CALIBRATION, **kw)
    out = capsys.readouterr().out
    assert out.count("running data:") == 3 and "Average optimized global pose mpjpe" in out and "joints error is" in out
    assert len(results) == 3 and len(opt) == 98 + 98 + 58 == len(est) == len(gt)
    ref_opt = np.concatenate([np.asarray(r[3]) for r in per_chunk])
    assert np.linalg.norm(np.asarray(opt) - ref_opt, axis=-1).mean() < 0.5e-3
    for k in results[0]:
        ref = np.mean([r[0][k] for r in per_chunk], axis=0)
        tol = 1e-9 if k.startswith("original") or k in ("aligned_original_mpjpe", "bone_length_aligned_original_mpjpe") else 0.5e-3
        np.testing.assert_allclose(summary[k], ref, rtol=0, atol=tol, err_msg=k)
    # host-side metrics and split batches give the same report
    torch.manual_seed(77)
    s2 = ws.optimize_directory(str(tmp_path), DEFAULT_CALIBRATION, chunks_per_batch=2, device_metrics=False, verbose=False, **kw)[0]
    for k in summary:
        np.testing.assert_allclose(s2[k], summary[k], rtol=0, atol=0.5e-3, err_msg=k)


def test_several_sequences_in_one_batch_match_sequence_by_sequence_runs(golden, tmp_path):
    """BASELINE configs[2] regime: the chunks of several sequences through the optimiser together; per-sequence reports
    must be those of running the sequences one after the other (same noise stream)."""
    import pickle
    import torch
    from globalegomocap_amd import whole_sequence as ws
    from helpers import sd_from_npz
    lt = golden("lbfgs_tiny")
    kw = dict(global_vae_path=sd_from_npz(lt, "global/"), local_vae_path=sd_from_npz(lt, "local/"), verbose=False)
    dirs = []
    for s, chunks in (("seqA", ((1, 100), (2, 60))), ("seqB", ((1, 100),)), ("seqC", ((1, 40), (2, 100), (3, 100)))):
        root = tmp_path / s
        root.mkdir()
        dirs.append(str(root))
        for i, n in chunks:
            d = root / ("chunk_%d" % i)
            d.mkdir()
            data = synth.make_sequence(n_frames=n, seed=100 * (ord(s[-1]) - ord("A")) + i)
            with open(d / "test_data.pkl", "wb") as f:
                pickle.dump(synth.reference_pickle_dict(data), f)          # as the reference's tool writes it
    torch.manual_seed(5)
    one_by_one = [ws.optimize_directory(d, DEFAULT_CALIBRATION, **kw) for d in dirs]
    torch.manual_seed(5)
    together = ws.optimize_sequences(dirs, DEFAULT_CALIBRATION, **kw)
    assert len(together) == 3
    for a, b in zip(one_by_one, together):
        assert len(a[1]) == len(b[1]) and len(a[3]) == len(b[3])
        assert np.linalg.norm(np.asarray(a[3]) - np.asarray(b[3]), axis=-1).mean() < 0.5e-3
        for k in a[0]:
            tol = 1e-9 if k.startswith("original") or k in ("aligned_original_mpjpe", "bone_length_aligned_original_mpjpe") else 0.5e-3
            np.testing.assert_allclose(b[0][k], a[0][k], rtol=0, atol=tol, err_msg=k)


def test_equal_chunks_take_the_vectorised_report_path_and_the_batch_pipeline(golden, tmp_path):
    """Equal 100-frame chunks written the reference's way (five keys, Fortran-ordered heat-maps, default protocol): all chunks'
    bookkeeping and error reports in one go; the same chunks as a PIPELINE of device calls (one or two chunks per call: the next
    call's files arrive and its noise is drawn while this one computes, three frame buffers rotate) must report what the
    per-chunk main() loop reports, and nothing may be written next to the data."""
    import pickle
    import torch
    from globalegomocap_amd import optimizer as gopt, whole_sequence as ws
    from helpers import sd_from_npz
    lt = golden("lbfgs_tiny")
    kw = dict(global_vae_path=sd_from_npz(lt, "global/"), local_vae_path=sd_from_npz(lt, "local/"))
    for i in range(5):
        d = tmp_path / ("chunk_%d" % i)
        d.mkdir()
        data = synth.make_sequence(n_frames=100, seed=70 + i, cam_jitter=(0.3, 0.002))
        with open(d / "test_data.pkl", "wb") as f:
            pickle.dump(synth.reference_pickle_dict(data), f)
    torch.manual_seed(9)
    per_chunk = [gopt.main(p, DEFAULT_CALIBRATION, 0.0, 0.0, 0.001, 0.01, 0.01, 0.01, final_smooth=True, **kw)
                 for p in ws.list_chunks(str(tmp_path))]
    runs = []
    for cpb in (None, 1, 2, None):
        torch.manual_seed(9)
        runs.append(ws.optimize_directory(str(tmp_path), DEFAULT_CALIBRATION, verbose=False, chunks_per_batch=cpb, **kw))
    assert sorted(os.listdir(tmp_path / "chunk_0")) == ["test_data.pkl"]
    ref_opt = np.concatenate([np.asarray(r[3]) for r in per_chunk])
    for summary, results, est, opt, gt in runs:
        assert len(results) == 5 and len(opt) == 5 * 98 == len(est) == len(gt)
        assert np.linalg.norm(np.asarray(opt) - ref_opt, axis=-1).mean() < 0.5e-3
        np.testing.assert_allclose(np.asarray(est), np.concatenate([np.asarray(r[1]) for r in per_chunk]), rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(np.asarray(gt), np.concatenate([np.asarray(r[4]) for r in per_chunk]), rtol=1e-9, atol=1e-12)
        for k in summary:
            ref = np.mean([r[0][k] for r in per_chunk], axis=0)
            tol = 1e-9 if k.startswith("original") or k in ("aligned_original_mpjpe", "bone_length_aligned_original_mpjpe") else 0.5e-3
            np.testing.assert_allclose(summary[k], ref, rtol=0, atol=tol, err_msg=k)
    # the same call twice: the same device inputs and noise, so the same results bit for bit
    assert np.array_equal(np.asarray(runs[0][3]), np.asarray(runs[3][3]))


@pytest.mark.parametrize("H,W,J", [(64, 64, 15), (48, 40, 13), (7, 5, 3), (1, 9, 2), (130, 3, 16)])
def test_heat_gather_kernels_against_numpy(H, W, J):
    """gem_heat_gather by itself: n payloads at ODD byte offsets of an image in HBM, C and Fortran order, float32 and float64, the
    path's 64 x 64 x 15 maps (compile-time geometry) and other shapes (run-time geometry, partial column tiles) -> [n,H,W,J] float32,
    bit for bit what numpy's reshape(order) + astype(float32) give."""
    import ctypes as C
    import torch
    from globalegomocap_amd import _capi
    lib = _capi.load_library()
    rng = np.random.default_rng(H * 1000 + W * 10 + J)
    n = 7
    for dt, np_dt in ((0, np.float32), (1, np.float64)):
        for fortran in (0, 1):
            arrays = [rng.standard_normal((H, W, J)).astype(np_dt) for _ in range(n)]
            blob, offs = bytearray(b"\x7f" * 5), []
            for k, a in enumerate(arrays):
                offs.append(len(blob))
                blob += a.tobytes(order="F" if fortran else "C") + b"\x01" * (1 + 2 * k)          # every alignment 0 .. 3 occurs
            blob += b"\0" * 16
            image = torch.frombuffer(bytearray(blob), dtype=torch.uint8).cuda()
            offs_d = torch.as_tensor(offs, dtype=torch.int64, device="cuda")
            out = torch.full((n, H, W, J), float("nan"), device="cuda")
            _capi.check(lib.gem_heat_gather(C.c_void_p(image.data_ptr()), len(blob) - 16, C.c_void_p(offs_d.data_ptr()), n, H, W, J, dt, fortran,
                                            C.c_void_p(out.data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream)), lib)
            expect = np.stack(arrays).astype(np.float32)
            assert np.array_equal(out.cpu().numpy().view(np.uint32), expect.view(np.uint32)), (dt, fortran)


def test_a_broken_chunk_in_a_later_batch_surfaces_and_leaves_the_pipeline_usable(golden, tmp_path):
    """Batches are in flight three deep (files arriving, on the device, being reported): a chunk without `camera_pose_list` in the
    THIRD batch must surface as the reference's KeyError (optimizer.py:322) after the earlier batches' device work has been
    waited for, and the next call must run as if nothing had happened."""
    import pickle
    import torch
    from globalegomocap_amd import whole_sequence as ws
    from helpers import sd_from_npz
    lt = golden("lbfgs_tiny")
    kw = dict(global_vae_path=sd_from_npz(lt, "global/"), local_vae_path=sd_from_npz(lt, "local/"))
    good, bad = tmp_path / "good", tmp_path / "bad"
    for root in (good, bad):
        for i in range(4):
            d = root / ("chunk_%d" % i)
            d.mkdir(parents=True)
            obj = synth.reference_pickle_dict(synth.make_sequence(n_frames=100, seed=90 + i))
            if root is bad and i == 2:
                del obj["camera_pose_list"]
            with open(d / "test_data.pkl", "wb") as f:
                pickle.dump(obj, f)
    torch.manual_seed(3)
    first = ws.optimize_directory(str(good), DEFAULT_CALIBRATION, verbose=False, chunks_per_batch=1, **kw)
    with pytest.raises(KeyError, match="camera_pose_list"):
        ws.optimize_directory(str(bad), DEFAULT_CALIBRATION, verbose=False, chunks_per_batch=1, **kw)
    with pytest.raises(FileNotFoundError):
        ws.optimize_directory(str(tmp_path / "nowhere"), DEFAULT_CALIBRATION, verbose=False, **kw)
    torch.manual_seed(3)
    again = ws.optimize_directory(str(good), DEFAULT_CALIBRATION, verbose=False, chunks_per_batch=1, **kw)
    assert np.array_equal(first[3], again[3]) and len(again[1]) == 4


# ------------------------------------------------------------------------------------------------ INTEGRATION.md, executed
def test_integration_md_ctypes_stub_runs_verbatim():
    """The reference-side ctypes stub of INTEGRATION.md section 2 (what a maintainer would paste into
    BodyPoseOptimizer.optimize_pose_seq_pytorch_LBFGS, /root/reference/optimizer.py:242-276), extracted from the markdown and
    executed VERBATIM against libgem_hip.so -- its own CDLL handle, no argtypes, its own struct mirror -- with stand-ins for the
    reference objects it reads (`self.fisheye_camera_model`, `self.kinematic_parents`, the weights, `state_dict` in ConvVAE module
    order).  Its result must be bitwise what WindowEngine.optimize_stage returns for the same window and noise."""
    import re
    import textwrap
    import types
    import torch
    from globalegomocap_amd import _capi, vae as V
    from globalegomocap_amd.camera import FisheyeCamera
    from globalegomocap_amd.engine import WindowEngine, energy_weights, LOCAL_STAGE
    from globalegomocap_amd.skeleton import KINEMATIC_PARENTS
    from globalegomocap_amd.vae_torch import MotionVAE
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    sec = text.split("## 2.", 1)[1]
    code = re.search(r"```python\n(.*?)```", sec, flags=re.S).group(1)
    assert "gem_optimize_stage" in code and "gem_load_vae" in code
    ns = {}
    exec("def stub(self, state_dict, relative_global_pose, heatmap_seq, GemLbfgsOpts):\n" + textwrap.indent(code, "    "), ns)
    FULL = V.VAEShape()
    cam = FisheyeCamera.from_json(DEFAULT_CALIBRATION)
    sd_np = V.structured_state_dict(FULL, 7, feature_offset=0.0)
    net = MotionVAE(FULL.latent_dim, FULL.seq_len, tuple(FULL.hidden))          # the reference's module layout: state_dict() in ConvVAE order, num_batches_tracked included
    net.load_state_dict({k: torch.as_tensor(np.asarray(v)) for k, v in sd_np.items()}, strict=False)
    state_dict = net.state_dict()
    assert any(not v.dtype.is_floating_point for v in state_dict.values())
    seq = synth.make_sequence(n_frames=10, seed=21, camera=cam)
    pose = np.asarray(seq["estimated_local_skeleton"], dtype=np.float32)
    heat = np.asarray(seq["heatmap_list"], dtype=np.float32)
    eng = WindowEngine(FULL, cam, max_windows=8)
    try:
        eng.load_vae(LOCAL_STAGE, sd_np)
        mb = eng.mean_bone_length(pose)
        me = types.SimpleNamespace(fisheye_camera_model=types.SimpleNamespace(fisheye_inverse_polynomial=np.asarray(cam.poly_w2c),
                                                                             img_center=np.array([cam.cx, cam.cy])),
                                   kinematic_parents=list(KINEMATIC_PARENTS), weight_3d=1e-6, smooth_weight=1e-5, bone_length_weight=1e-2,
                                   vae_weight=0.0, reproj_weight=1e-2, mean_bone_length=mb.reshape(1, 15).contiguous())
        cwd = os.getcwd()
        os.chdir(root)              # (the stub opens the library by its path relative to the repository)
        try:
            torch.manual_seed(5)
            out = ns["stub"](me, state_dict, pose, heat, _capi.GemLbfgsOpts)
        finally:
            os.chdir(cwd)
        torch.manual_seed(5)
        eps = torch.randn(1, 2048)
        ref, stats = eng.optimize_stage(LOCAL_STAGE, pose[None], mb, eps, energy_weights(1e-6, 1e-5, 1e-2, 0.0, 1e-2), heat, np.zeros(1, np.int32))
        assert int(stats.cpu().numpy()[0, 3]) == 1 and int(stats.cpu().numpy()[0, 1]) > 5
        assert out.shape == (10, 15, 3) and out.dtype == np.float32
        assert np.array_equal(out, ref[0].cpu().numpy())
        assert np.abs(out - pose).max() > 1e-4          # (the stage moved the pose)
    finally:
        eng.close()


def test_main_returns_the_reference_container_types(golden, tmp_path):
    """optimizer.main's 5-tuple like /root/reference/optimizer.py:425-450,506-507: merge_batches builds LISTS of [15,3] frames; only
    final_smooth=True turns final_optimized_seq into an ndarray (gaussian_filter1d)."""
    import pickle
    import torch
    from collections import OrderedDict
    from globalegomocap_amd import optimizer as gopt
    from helpers import sd_from_npz
    g, lt = golden("pipeline_tiny"), golden("lbfgs_tiny")
    data = synth.make_sequence(n_frames=100, seed=int(g["seq_seed"]))
    d = tmp_path / "chunk0"
    d.mkdir()
    with open(d / "test_data.pkl", "wb") as f:
        pickle.dump({k: data[k] for k in ("estimated_local_skeleton", "gt_global_skeleton", "camera_pose_list", "heatmap_list")}, f)
    args = (str(d), DEFAULT_CALIBRATION, 0.0, 0.0, float(g["smooth"]), 0.01, float(g["weight_3d"]), 0.01)
    for smooth in (False, True):
        torch.manual_seed(int(g["eps_seed"]))
        res = gopt.main(*args, final_smooth=smooth, global_vae_path=sd_from_npz(lt, "global/"), local_vae_path=sd_from_npz(lt, "local/"),
                        eps=torch.randn(24, 32))
        errors, est, mid, opt, gt = res
        assert isinstance(errors, OrderedDict) and len(errors) == 18
        for name, seq_ in (("final_estimated_seq", est), ("mid_local_pose_seq", mid), ("final_gt_seq", gt)):
            assert isinstance(seq_, list) and len(seq_) == 98 and np.asarray(seq_[0]).shape == (15, 3), name
        if smooth:
            assert isinstance(opt, np.ndarray) and opt.shape == (98, 15, 3) and opt.dtype == np.float64
        else:
            assert isinstance(opt, list) and len(opt) == 98 and opt[0].shape == (15, 3) and opt[0].dtype == np.float64


def test_save_pose_writes_the_reference_result_pickle(golden, tmp_path, monkeypatch):
    """main(..., save_pose=True) (/root/reference/optimizer.py:469-483): `out/<dataset>/<sequence>/result_pose.pkl` under the working
    directory with the keys estimated_pose / optimized_pose / mid_optimized_pose / gt_pose, in the reference's containers, holding
    the sequences the call returns (mid_optimized_pose is the GLOBAL stage-one sequence the error report's mid_* entries use)."""
    import pickle
    import torch
    from globalegomocap_amd import optimizer as gopt
    from globalegomocap_amd.errors import mpjpe
    from helpers import sd_from_npz
    g, lt = golden("pipeline_tiny"), golden("lbfgs_tiny")
    data = synth.make_sequence(n_frames=100, seed=int(g["seq_seed"]))
    d = tmp_path / "studio-x" / "chunk_7"
    d.mkdir(parents=True)
    with open(d / "test_data.pkl", "wb") as f:
        pickle.dump(synth.reference_pickle_dict(data), f)
    monkeypatch.chdir(tmp_path)
    for smooth in (True, False):
        errors, est, mid, opt, gt = gopt.main(str(d), DEFAULT_CALIBRATION, 0.0, 0.0, float(g["smooth"]), 0.01, float(g["weight_3d"]), 0.01,
                                             final_smooth=smooth, save_pose=True, global_vae_path=sd_from_npz(lt, "global/"),
                                             local_vae_path=sd_from_npz(lt, "local/"), eps=torch.randn(24, 32, generator=torch.Generator().manual_seed(5)))
        out = tmp_path / "out" / "studio-x" / "chunk_7" / "result_pose.pkl"
        assert out.exists()
        with open(out, "rb") as f:
            saved = pickle.load(f)
        assert list(saved) == ["estimated_pose", "optimized_pose", "mid_optimized_pose", "gt_pose"]
        for k in ("estimated_pose", "mid_optimized_pose", "gt_pose"):
            assert isinstance(saved[k], list) and len(saved[k]) == 98 and np.asarray(saved[k][0]).shape == (15, 3), k
        assert isinstance(saved["optimized_pose"], np.ndarray if smooth else list)
        assert np.array_equal(np.asarray(saved["estimated_pose"]), np.asarray(est)) and np.array_equal(np.asarray(saved["gt_pose"]), np.asarray(gt))
        assert np.array_equal(np.asarray(saved["optimized_pose"]), np.asarray(opt))
        assert abs(mpjpe(np.asarray(saved["mid_optimized_pose"]), np.asarray(saved["gt_pose"])) - errors["mid_global_mpjpe"]) < 1e-12
        assert abs(mpjpe(np.asarray(saved["optimized_pose"]), np.asarray(saved["gt_pose"])) - errors["optimized_global_mpjpe"]) < 1e-12
        out.unlink()


def test_integration_md_chunk_reader_stub_runs_verbatim(tmp_path):
    """The chunk-reader stub of INTEGRATION.md section 2 (what replaces pickle.load + np.asarray(heatmap_list) + .float() at
    /root/reference/optimizer.py:315-324,248), extracted from the markdown and executed VERBATIM -- its own CDLL handle and struct
    mirror -- on a file written the way the reference's tool writes it; its `heat` and `est` must be exactly what the reference's
    three lines produce."""
    import ctypes as C
    import pickle
    import re
    import torch
    md = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", md, flags=re.S)
    stub = next(b for b in blocks if "gem_chunk_open" in b)
    head = next(b for b in blocks if "lib = C.CDLL(" in b).split("class Cfg")[0]          # the CDLL line of the first stub
    data = synth.make_sequence(n_frames=40, seed=12)
    d = tmp_path / "chunk_0"
    d.mkdir()
    with open(d / "test_data.pkl", "wb") as f:
        pickle.dump(synth.reference_pickle_dict(data), f)
    env = {"C": C, "np": np, "torch": torch, "path": str(d / "test_data.pkl")}
    cwd = os.getcwd()
    os.chdir(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    try:
        exec(head, env)
        exec(stub, env)
    finally:
        os.chdir(cwd)
    torch.cuda.synchronize()
    with open(d / "test_data.pkl", "rb") as f:
        ref = pickle.load(f)
    assert np.array_equal(env["heat"].cpu().numpy(), torch.from_numpy(np.asarray(ref["heatmap_list"])).float().numpy())
    assert np.array_equal(env["est"], np.asarray(ref["estimated_local_skeleton"]))
    assert env["n"] == 40 and list(env["info"])[:4] == [40, 3, 0, 1]          # 40 arrays, 3-D, float32, Fortran order


def test_integration_md_reporting_stub_runs_verbatim(engine, golden):
    """The second ctypes block of INTEGRATION.md section 2 (the reporting call a maintainer would put behind
    /root/reference/optimize_whole_sequence.py's calculate_errors, calculate_errors.py:114-179), executed verbatim against the library
    -- its own CDLL handle, no argtypes -- on the sequences of the reference-generated golden run; its 17 + 15 numbers must be the
    engine's own `calculate_errors_device` bit for bit and the reference's golden error dict (same tolerances as the test above)."""
    import ctypes as C
    import re
    import types
    import torch
    from globalegomocap_amd.skeleton import mean_bone_length_mm
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sec = open(os.path.join(root, "INTEGRATION.md")).read().split("## 2.", 1)[1]
    code = next(b for b in re.findall(r"```python\n(.*?)```", sec, flags=re.S) if "gem_calculate_errors" in b)
    g = golden("pipeline_tiny")
    cams = np.asarray(synth.make_sequence(n_frames=100, seed=int(g["seq_seed"]), with_heatmaps=False)["camera_pose_list"])
    lib = C.CDLL(os.path.join(root, "globalegomocap_amd", "_lib", "libgem_hip.so"))
    for tag in ("smooth", "raw"):
        mid = g["mid_local_" + tag] + cams[:98, :3, 3][:, None, :]
        est, mid, opt, gt = (torch.as_tensor(np.ascontiguousarray(x), dtype=torch.float64, device="cuda")
                             for x in (g["est_" + tag], mid, g["opt_" + tag], g["gt_" + tag]))
        ns = {"C": C, "torch": torch, "lib": lib, "h": engine._h, "est": est, "mid": mid, "opt": opt, "gt": gt,
              "skeleton_model": types.SimpleNamespace(bone_length=[float(v) for v in mean_bone_length_mm()]),
              "stream": C.c_void_p(torch.cuda.current_stream().cuda_stream)}
        exec(code, ns)
        torch.cuda.synchronize()
        assert ns["rc"] == 0
        assert torch.equal(ns["out"], engine.calculate_errors_device(est, mid, opt, gt))
        rep = ns["out"].cpu().numpy()
        for i, k in enumerate(engine.ERROR_KEYS):
            tol = dict(rtol=1e-5, atol=1e-7) if "mid" in k else dict(rtol=1e-9, atol=1e-12)
            np.testing.assert_allclose(rep[i], g["err_%s/%s" % (tag, k)], err_msg=k, **tol)
        np.testing.assert_allclose(rep[17:], g["err_%s/joints_error" % tag], rtol=1e-9, atol=1e-12)


def test_chunk_pickles_reach_the_device_without_passing_through_python_objects(tmp_path, monkeypatch):
    """load_chunk(path, device): the library interprets the pickle (gem_chunk_open), the FILE goes to pinned memory and on to the
    device in slices (gem_file_stage), and one kernel there picks the 100 arrays out -- undoing the Fortran order of
    scipy.io.loadmat's arrays and rounding float64 to float32 on the way (gem_heat_gather).  The cases are built the reference's
    way (process_test_data.py:65-67,149-157): heat-maps through savemat / loadmat, float32 AND float64, five keys, every protocol
    that stores raw bytes; C-ordered lists (what np.ascontiguousarray'd data give) as well.  Files outside the library reader's
    subset (protocol 2, ragged lists) are stacked on the host.  Every path must deliver exactly what
    `torch.from_numpy(np.asarray(pickle.load(f)['heatmap_list'])).float()` holds (optimizer.py:324,248)."""
    import pickle
    import scipy.io as sio
    import torch
    from globalegomocap_amd import _capi, whole_sequence as ws
    rng = np.random.default_rng(3)
    n = 100
    small = {"gt_global_skeleton": list(rng.random((n, 15, 3))), "estimated_global_skeleton": list(rng.random((n, 15, 3))),
             "estimated_local_skeleton": list(rng.random((n, 15, 3))), "camera_pose_list": list(rng.random((n, 4, 4)))}
    lib = _capi.load_library()
    calls = {"native": 0}
    orig = lib.gem_heat_gather

    class Counting:                                   # (ctypes function pointers cannot be monkeypatched in place)
        def __getattr__(self, name):
            if name == "gem_heat_gather":
                def f(*a):
                    rc = orig(*a)
                    calls["native"] += rc == 0
                    return rc
                return f
            return getattr(lib, name)
    monkeypatch.setattr(_capi, "load_library", lambda path=None: Counting())

    def from_mat(a, k):
        sio.savemat(str(tmp_path / ("h%d.mat" % k)), {"heatmap": a})
        return sio.loadmat(str(tmp_path / ("h%d.mat" % k)))["heatmap"]
    base32 = rng.random((n, 64, 64, 15), dtype=np.float32)
    base32[0, :4, :4, 0] = [1e-40, -1e-42, 0.0, -0.0]                 # (denormals and signed zeros travel too)
    base64 = rng.random((n, 64, 64, 15)) * 1e-3
    base64[0, 0, :8, 1] = [1e-40, 1e-46, 1.0000000596046448, 1.00000017881393433, 3.5e38, -3.5e38, np.inf, np.nan]      # rounding cases of the f64 -> f32 cast
    dev = torch.device("cuda:0")
    cases = [("F32", base32, "mat", None, 1), ("F64", base64, "mat", None, 1), ("F32", base32, "mat", 3, 1), ("F64", base64, "mat", 5, 1),
             ("C32", base32, "C", 4, 1), ("C64", base64, "C", 3, 1), ("C32", base32, "C", 5, 1), ("F32", base32, "mat", 2, 0),
             ("ragged", base32, "ragged", 4, 0)]
    for ci, (tag, data, order, proto, native) in enumerate(cases):
        d = tmp_path / ("c_%d" % ci)
        d.mkdir()
        if order == "mat":
            hl = [from_mat(h, ci) for h in data[:6]] + [np.asfortranarray(h) for h in data[6:]]      # (loadmat for the first few, its layout for the rest)
            assert hl[0].flags.f_contiguous and not hl[0].flags.c_contiguous and hl[0].dtype == data.dtype
        else:
            hl = [h.copy() for h in data]
        expect = torch.from_numpy(np.asarray(hl)).float().numpy()
        if order == "ragged":
            hl[-1] = np.zeros((64, 64, 15), dtype=np.float64)          # one array of another type: the list is stacked on the host
            expect[-1] = 0.0
        obj = dict(small, heatmap_list=hl)
        with open(d / "test_data.pkl", "wb") as f:
            pickle.dump(obj, f) if proto is None else pickle.dump(obj, f, protocol=proto)
        calls["native"] = 0
        dest = torch.full((n, 64, 64, 15), -1.0, dtype=torch.float32, device=dev)
        torch.cuda.synchronize()                       # (the reader copies on its own stream)
        c = ws.load_chunk(str(d), device=dev, dest=dest)
        c["heat_ready"].synchronize()
        assert calls["native"] == native, (tag, proto, calls)
        assert c["heat"].data_ptr() == dest.data_ptr()
        got = c["heat"].cpu().numpy()
        assert np.array_equal(got.view(np.uint32), expect.view(np.uint32)), (tag, proto)          # bit for bit (NaN, -0.0 and denormals included)
        for k, name in (("est_local", "estimated_local_skeleton"), ("gt", "gt_global_skeleton"), ("cams", "camera_pose_list")):
            assert np.array_equal(c[k], np.asarray(small[name])), (tag, k)
        # a second and third chunk through the same reader thread's buffers (both staging buffers get reused)
        for _ in range(2):
            c2 = ws.load_chunk(str(d), device=dev)
            c2["heat_ready"].synchronize()
            assert np.array_equal(c2["heat"].cpu().numpy().view(np.uint32), expect.view(np.uint32)), (tag, proto)
