"""Run-to-run determinism of the kernels that share a SIMD with MFMAs, and the hardware hazard behind the one failure this code base
has seen ("packed fp32"): on gfx950 a packed fp32 VALU instruction whose op_sel takes the high half of its second
source for the low result returns wrong lanes 48-63 while another wavefront's v_mfma_f32_16x16x32_bf16 runs on the same SIMD.  The
library is built without packed fp32 arithmetic (tests/test_host_cpu.py checks the disassembly); these tests check the symptom --
repeated launches of the kernels at full occupancy must agree bit for bit -- and keep the stand-alone reproducer alive."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from globalegomocap_amd import synth
from globalegomocap_amd.camera import FisheyeCamera, DEFAULT_CALIBRATION
from helpers import FULL

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("windows,precision", [(8192, "bf16"), (1536, "bf16"), (240, "bf16"), (1536, "f32"), (240, "f32")])
def test_repeated_evaluations_are_bitwise_identical(windows, precision):
    """TEN evaluations (energy, its five parts, dE/dz, decoded pose) of the same latent points with the reprojection term on: the
    bf16 tail at two workgroups per CU (8192 windows: one workgroup in its matrix layers while its neighbour computes energy terms
    -- the constellation in which ~1 % of the windows differed from launch to launch with packed fp32 arithmetic in the energy
    terms), at three / one windows per workgroup, and the fp32 paths (batched narrow layers + energy_kernel at 1536, the fused
    one-window tail at 240).  /root/reference/optimizer.py:139-149,226-240 evaluated twice gives the same numbers; so must this."""
    import torch
    from globalegomocap_amd import vae as V
    from globalegomocap_amd.engine import WindowEngine, energy_weights
    cam = FisheyeCamera.from_json(DEFAULT_CALIBRATION)
    sd = V.structured_state_dict(FULL, 7, feature_offset=0.0, signal_offset=1.0)
    eng = WindowEngine(FULL, cam, max_windows=windows)
    try:
        eng.load_vae(0, sd)
        eng.set_precision(precision)
        n_frames = 3000
        seq = synth.make_sequence_device(n_frames, seed=303, device=eng.device, cam_jitter=(0.3, 0.002))
        starts = np.random.default_rng(303).integers(0, n_frames - 10, windows).astype(np.int32)
        f0 = torch.as_tensor(starts, device=eng.device)
        pose = seq["est_local"][f0.long()[:, None] + torch.arange(10, device=eng.device)[None]].contiguous()
        mb = eng.mean_bone_length(seq["est_local"][:100]).reshape(1, 15).expand(windows, 15).contiguous()
        eps = torch.randn(windows, FULL.latent_dim, generator=torch.Generator().manual_seed(1)).to(eng.device)
        _, _, z = eng.encode(0, pose.reshape(windows, 10, 45), eps)
        w = energy_weights(1e-6, 1e-5, 1e-2, 0.0, 1e-2)
        ref = None
        for rep in range(10):
            cur = [t.clone() for t in eng.energy_grad(0, z, pose, mb, w, seq["heat"], f0)]
            torch.cuda.synchronize()
            if ref is None:
                ref = cur
                assert all(bool(torch.isfinite(t).all()) for t in ref)
                continue
            for name, a, b in zip(("E", "parts", "dz", "X"), ref, cur):
                if not torch.equal(a, b):
                    bad = (a != b).reshape(windows, -1).any(dim=1).nonzero().flatten().cpu().numpy()
                    raise AssertionError("evaluation %d: %s differs from the first on %d of %d windows (first %s)" % (rep, name, len(bad), windows, bad[:8]))
    finally:
        eng.close()


def test_bf16_tail_instantiations_agree_over_8192_windows_ten_times(monkeypatch):
    """The bf16 tail with 2 row tiles per workgroup (3 windows, two workgroups per CU) against 5 row tiles (8 windows, one energy
    trip per window group more): the same 8192 windows evaluated TEN times by each, alternating -- energies, parts and poses must
    agree bit for bit between the two and from repeat to repeat, dE/dz included (round 5 saw one rounding of one window's gradient
    fall the other way: the compiler's fused-multiply-add choice under -ffp-contract=fast differed between the instantiations;
    the library is built with -ffp-contract=on since round 6).  What is seen is recorded with the card's
    identity either way."""
    import torch
    from globalegomocap_amd import vae as V
    from globalegomocap_amd.engine import WindowEngine, energy_weights
    from helpers import record_observation
    windows = 8192
    cam = FisheyeCamera.from_json(DEFAULT_CALIBRATION)
    sd = V.structured_state_dict(FULL, 7, feature_offset=0.0, signal_offset=1.0)
    monkeypatch.setenv("GEM_DEV", "1")
    monkeypatch.setenv("GEM_TAIL16", "1")
    eng = WindowEngine(FULL, cam, max_windows=windows)
    try:
        eng.load_vae(0, sd)
        eng.set_precision("bf16")
        n_frames = 3000
        seq = synth.make_sequence_device(n_frames, seed=404, device=eng.device, cam_jitter=(0.3, 0.002))
        starts = np.random.default_rng(404).integers(0, n_frames - 10, windows).astype(np.int32)
        f0 = torch.as_tensor(starts, device=eng.device)
        pose = seq["est_local"][f0.long()[:, None] + torch.arange(10, device=eng.device)[None]].contiguous()
        mb = eng.mean_bone_length(seq["est_local"][:100]).reshape(1, 15).expand(windows, 15).contiguous()
        eps = torch.randn(windows, FULL.latent_dim, generator=torch.Generator().manual_seed(2)).to(eng.device)
        _, _, z = eng.encode(0, pose.reshape(windows, 10, 45), eps)
        w = energy_weights(1e-6, 1e-5, 1e-2, 0.0, 1e-2)
        first, per_repeat, self_repeats = {}, [], {2: True, 5: True}
        for rep in range(10):
            cur = {}
            for nrt in (2, 5):
                monkeypatch.setenv("GEM_TAIL16_NRT", str(nrt))
                cur[nrt] = [t.clone() for t in eng.energy_grad(0, z, pose, mb, w, seq["heat"], f0)]
                torch.cuda.synchronize()
                if rep == 0:
                    first[nrt] = cur[nrt]
                else:
                    self_repeats[nrt] &= all(torch.equal(a, b) for a, b in zip(first[nrt], cur[nrt]))
            for k in (0, 1, 3):                                   # energy, parts, decoded pose: bit for bit between the instantiations
                assert torch.equal(cur[2][k], cur[5][k]), (rep, ("E", "parts", "dz", "X")[k])
            d2, d5 = cur[2][2], cur[5][2]
            bad = (d2 != d5).any(dim=1).nonzero().flatten().cpu().numpy()
            per_repeat.append({"windows_differing": int(bad.size), "indices": bad[:8].tolist(),
                               "max_abs_over_largest": float((d2 - d5).abs().max() / d5.abs().max())})
        record_observation("bf16_tail_nrt2_vs_nrt5_8192", {"windows": windows, "repeats": per_repeat, "each_instantiation_repeats_bitwise": self_repeats,
                                                          "bitwise": all(r["windows_differing"] == 0 for r in per_repeat)})
        assert self_repeats[2] and self_repeats[5], self_repeats
        for r in per_repeat:
            assert r["windows_differing"] == 0, r
    finally:
        eng.close()


def test_training_steps_repeat_bitwise():
    """Two trainers from the same state, three steps of 1024 windows each (the batch at which the step's matrix products fill the
    chip, elementwise / BatchNorm kernels run beside them on the same stream): parameters, moments and statistics bit for bit."""
    from globalegomocap_amd.vae_train import VAETrainer, initial_state_dict
    B = 1024
    init = initial_state_dict(FULL, 5)
    poses = synth.make_training_windows(3 * B, FULL.seq_len, 4).reshape(3, B, FULL.seq_len, 45)
    eps = np.random.default_rng(3).standard_normal((3, B, FULL.latent_dim)).astype(np.float32)
    runs = []
    for _ in range(2):
        tr = VAETrainer(FULL, batch_size=B, lr=1e-3, weight_decay=1e-5, state_dict=init)
        try:
            losses = [tr.step(poses[s], 0.01, eps=eps[s], keep_gradients=False) for s in range(3)]
            runs.append((losses, [tr._down(what).copy() for what in (0, 2, 3, 4)]))
        finally:
            tr.close()
    assert runs[0][0] == runs[1][0]
    for a, b in zip(runs[0][1], runs[1][1]):
        assert np.array_equal(a, b)


def test_packed_fp32_reproducer():
    """tools/slp_hazard/pk_mfma_repro.hip, compiled here with hipcc and run: thirteen packed instruction forms and (round 6) four more
    VALU forms of the bf16 tail's energy phase -- v_cvt_pk_bf16_f32, v_mov_b64, a DPP row shift, v_add_f64 -- x {no matrix waves,
    three MFMA shapes}.  Asserted: the program runs, the control without matrix waves is clean, and the forms the library could
    still contain after the build's -packed-fp32-ops (plain v_pk_mov_b32) are clean beside every MFMA shape.  Whether the hazard shows
    in a given run (it did on every MI355X box used in round 5: v_pk_{fma,mul,add}_f32 with op_sel:[0,1..] beside
    v_mfma_f32_16x16x32_bf16, lanes 48-63 only) is printed, not asserted: a fixed part would be good news."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc on this box")
    import tempfile
    tmp = tempfile.mkdtemp(prefix="pk_repro_")
    try:
        exe = os.path.join(tmp, "pk_mfma_repro")
        subprocess.run([hipcc, "-w", "--offload-arch=gfx950", "-O3", "-std=c++17", "-o", exe, os.path.join(ROOT, "tools", "slp_hazard", "pk_mfma_repro.hip")],
                       check=True, timeout=600)
        r = subprocess.run([exe, "5000"], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        print(r.stdout)
        section, seen = None, {}
        for line in r.stdout.splitlines():
            if line.startswith("-- matrix waves:"):
                section = line.split(":")[1].split(";")[0].strip()
            elif "wrong low halves" in line:
                form = line.split("wrong low halves")[0].strip()
                lo, hi = int(line.split("wrong low halves")[1].split(",")[0]), int(line.split("wrong high halves")[1].split(",")[0])
                seen[(section, form)] = lo + hi
                mask = int(line.rsplit("lanes", 1)[1].strip(), 16)
                assert mask & ~0xFFFF000000000000 == 0, ("a wrong result outside lanes 48-63", line)
        assert len(seen) == 4 * 17
        assert all(v == 0 for (sec, _), v in seen.items() if sec == "no matrix waves"), seen
        assert all(v == 0 for (_, form), v in seen.items() if form.startswith("v_pk_mov_b32")), seen
        assert all(v == 0 for (_, form), v in seen.items() if "op_sel" not in form), seen
        assert r.stdout.strip().splitlines()[-1] in ("PACKED_FP32_HAZARD_SEEN", "PACKED_FP32_CLEAN")
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
