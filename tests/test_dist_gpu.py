"""RCCL smoke test of the one collective of the path on the single GPU of the test box: a world of one rank on
backend "nccl" (= RCCL on ROCm) through the same helper `bench.py --gpus N` uses.  Ragged shards and real
multi-rank exchange are covered on CPU with gloo (tests/test_dist_cpu.py)."""
import os
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, %r)
    import torch, torch.distributed as dist
    from globalegomocap_amd.dist import all_gather_windows, optimize_sharded
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%%d" %% int(sys.argv[1]), rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    x = torch.randn(7, 10, 15, 3, device="cuda", dtype=torch.float64)
    y = all_gather_windows(x, 7)
    assert y.is_cuda and y.dtype == torch.float64 and torch.equal(x, y)
    z = optimize_sharded(lambda lo, hi, f0, f1: x[lo:hi].float(), list(range(0, 56, 8)))
    assert torch.equal(z, x.float())
    dist.barrier()
    dist.destroy_process_group()
    print("RCCL_OK")
""") % ROOT


def test_all_gather_over_rccl_single_rank(tmp_path):
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    p = tmp_path / "rccl_one.py"
    p.write_text(SCRIPT)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, str(p), str(port)], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and "RCCL_OK" in r.stdout, r.stdout + r.stderr


def _run_bench(args, env_extra, timeout=900):
    import json
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **env_extra)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=timeout, env=env)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.parametrize("workload,windows,precision", [("configs4", 150, "f32"), ("configs4", 150, "bf16"), ("configs3", 130, "bf16")])
def test_two_rank_rehearsal_of_the_sharded_bench_legs_is_bitwise_the_emulated_shards(tmp_path, workload, windows, precision):
    """`bench.py --gpus 2 --workload configs3|configs4` with two REAL ranks (fresh child processes under torch.distributed.run,
    gloo, both on the one card of the test box: GEM_BENCH_REHEARSAL=1) against the same two shards run one after the other by ONE
    process without any collective (--emulate-ranks 2).  Windows do not interact, so sharding, per-rank frame sets (block-cyclic
    blocks + halo for the stream), the all-gather and the index scatter must not change a bit of the gathered [n,10,15,3]
    poses (nor of the merged + smoothed sequence of the stream).  (A ONE-batch run of all windows is a different computation --
    the split-K cut of the GEMMs follows the batch size, and L-BFGS trajectories that differ by rounding part at the kinks of the
    energy, DESIGN.md 5.1 -- so it agrees to a fraction of a millimetre, not bitwise: measured 0.16 mm mean in fp32.)"""
    import numpy as np
    common = ["--workload", workload, "--windows", str(windows), "--block", "16", "--steps", "2", "--warmup", "1", "--vae", "structured",
              "--precision", precision, "--cpu-windows", "0"]
    d2, de, d1 = (str(tmp_path / n) for n in ("two.npz", "emu.npz", "one.npz"))
    two = _run_bench(["--gpus", "2"] + common + ["--dump", d2], {"GEM_BENCH_REHEARSAL": "1"})
    emu = _run_bench(["--gpus", "1"] + common + ["--emulate-ranks", "2", "--dump", de], {})
    one = _run_bench(["--gpus", "1"] + common + ["--dump", d1], {})
    assert two["config"]["ranks"] == 2 and two["n_gpus"] == 2 and two["config"]["backend"] == "gloo" and two["scaling"] == "strong"
    assert emu["config"]["ranks"] == 1 and emu["config"]["emulated_ranks"] == 2
    assert two["all_finished"] and emu["all_finished"] and one["all_finished"]
    a, b, c = np.load(d2), np.load(de), np.load(d1)
    assert a["glob"].shape == (windows, 10, 15, 3)
    assert np.array_equal(a["glob"], b["glob"]) and np.array_equal(a["merged"], b["merged"])
    if workload == "configs4":
        assert a["merged"].shape == (8 * windows + 2, 15, 3)
        assert two["graph"]["replays"] >= 2
        # one batch of all windows: same windows, another split-K cut -> rounding-level (fp32) / bf16-level differences
        d = np.linalg.norm(a["glob"] - c["glob"], axis=-1).mean()
        assert d < (0.5e-3 if precision == "f32" else 8e-3), d          # (bf16 on the structured VAEs: DESIGN.md 5.1)


def test_default_two_rank_line_carries_the_strong_scaling_legs():
    """`bench.py --gpus 2` with default flags (what the driver runs with N > 1): the weak-scaling `value` of BASELINE configs[1]
    PLUS, as side records of the same JSON line, configs[3] and configs[4] as ONE job each sharded over the two ranks (scaling
    "strong", ranks 2, the per-rank evaluation sums = the load-imbalance term of SURVEY.md 8e).  Two real ranks on the one card
    of the test box over gloo (GEM_BENCH_REHEARSAL=1; the rehearsal takes a sixteenth of the windows)."""
    line = _run_bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--vae", "structured", "--cpu-windows", "0", "--no-profile"],
                      {"GEM_BENCH_REHEARSAL": "1"}, timeout=1500)
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["config"]["ranks"] == 2 and line["value"] > 0
    for key, n_windows in (("configs3", 65536 // 16), ("configs4", 12499 // 16)):
        rec = line[key]
        assert rec is not None and rec["scaling"] == "strong" and rec["n_gpus"] == 2 and rec["config"]["ranks"] == 2, (key, rec)
        assert rec["config"]["windows_total"] == n_windows and rec["all_finished"] and rec["value"] > 0, (key, rec)
        ev = rec["evaluations_per_rank"]
        assert len(ev["sum"]) == 2 and min(ev["sum"]) > 0 and 1.0 <= ev["max_over_mean"] < 1.5, (key, ev)
    assert line["configs4"]["graph"]["replays"] >= 1


@pytest.fixture(scope="module")
def fitted_weights_cache(tmp_path_factory):
    return str(tmp_path_factory.mktemp("bench_weights") / "vae_cache.pt")


@pytest.mark.parametrize("workload,windows,per_rank_max,imbalance", [("configs3", 65536, 8192, 1.1), ("configs4", 12499, 1568, 1.25)])
def test_full_size_partitions_emulated_8_ranks(fitted_weights_cache, workload, windows, per_rank_max, imbalance):
    """BASELINE configs[3] (65 536 windows, contiguous shards of 8192) and configs[4] (the 12 499 overlapping windows of ONE
    100 000-frame stream, block-cyclic shards + halo frames, hipGraph replay, gather by index, merge + smoothing) at their FULL
    size under the 8-way partition of the multi-GPU legs -- the eight shards run one after the other on the one card of the test
    box, each exactly as its rank would run it (`bench.py --emulate-ranks 8`; one shard resident at a time).  Windows are
    independent (/root/reference/optimizer.py:370), the merge is optimizer.py:425-437: every window must finish, the gathered
    array must be in window order, the optimisation must improve the MPJPE, and the per-rank work (evaluations) and time must be
    balanced -- the only thing that can cost the 8-GPU scaling."""
    line = _run_bench(["--gpus", "1", "--workload", workload, "--emulate-ranks", "8", "--steps", "3", "--warmup", "2", "--cpu-windows", "0",
                       "--weights-cache", fitted_weights_cache], {}, timeout=1500)
    cfg, part = line["config"], line["partition"]
    assert cfg["windows_total"] == windows and cfg["emulated_ranks"] == 8 and cfg["ranks"] == 1 and cfg["precision"] == "bf16"
    assert cfg["windows_per_rank_max"] == per_rank_max
    assert line["all_finished"] and part["gathered_order_is_arange"]
    pr = part["per_rank"]
    assert [p["rank"] for p in pr] == list(range(8)) and sum(p["windows"] for p in pr) == windows
    assert all(p["evaluations"] > 20 * p["windows"] and p["ms_per_step"] > 0 for p in pr), pr
    assert 1.0 <= part["evaluations_per_rank"]["max_over_mean"] < imbalance, part
    assert 1.0 <= part["time_per_rank"]["max_over_mean"] < imbalance, part
    assert line["mpjpe_optimised_mm"] < 0.6 * line["mpjpe_input_mm"], (line["mpjpe_input_mm"], line["mpjpe_optimised_mm"])
    proj = part["projected_8gpu_windows_per_s"]
    assert proj > 5.5 * line["value"] and "PROJECTION" in part["projection_note"]          # (8 x one card, less the imbalance and the merge)
    if workload == "configs4":
        held = sum(p["frames_held"] for p in pr)
        assert 100002 <= held <= 100002 + 2 * (windows // 8 + 1), held          # frames stored once + the 2-frame halos between blocks of 8
        assert line["graph"]["replays"] >= 8
    else:
        assert all(p["frames_held"] == 8 * (p["windows"] - 1) + 10 for p in pr)


def test_block_cyclic_shards_balance_a_recording_with_quiet_and_busy_stretches(fitted_weights_cache):
    """windows are independent (/root/reference/optimizer.py:370) but not equally expensive -- a window whose L-BFGS
    leaves early (optimizer.py:261-270) costs a fraction of one that runs to its evaluation limit.  `synth.activity_profile` makes a
    stream that alternates between quiet stretches (still wearer, blank heat-maps: the local stage leaves after ~1 evaluation, the
    global one after ~9) and busy ones (31 + 16), 500-5000 frames each.  On 3124 windows (25 000 frames) and eight emulated ranks
    CONTIGUOUS shards are about one stretch each -- the evaluations per rank must be visibly unbalanced, or the data would prove
    nothing -- and the block-cyclic shards bench.py's multi-GPU leg uses (blocks of 8 windows) must even them out."""
    common = ["--gpus", "1", "--workload", "configs4", "--windows", "3124", "--activity", "9", "--emulate-ranks", "8", "--steps", "2", "--warmup", "1",
              "--cpu-windows", "0", "--weights-cache", fitted_weights_cache]
    contiguous = _run_bench(common + ["--block", "0"], {}, timeout=1500)
    cyclic = _run_bench(common, {}, timeout=1500)
    for line in (contiguous, cyclic):
        assert line["all_finished"] and line["partition"]["gathered_order_is_arange"] and line["config"]["activity_profile"]
    ev_c, ev_b = contiguous["evaluations_per_rank"], cyclic["evaluations_per_rank"]
    print("contiguous:", ev_c, contiguous["partition"]["time_per_rank"], "| blocks of 8:", ev_b, cyclic["partition"]["time_per_rank"])
    assert ev_c["max_over_mean"] >= 1.3, ev_c                     # the test data really is heterogeneous
    assert ev_b["max_over_mean"] <= 1.10, ev_b
    assert cyclic["partition"]["time_per_rank"]["max_over_mean"] <= 1.10
    # the same windows either way: the same total work, the same accuracy
    assert abs(sum(ev_c["sum"]) - sum(ev_b["sum"])) <= 0.02 * sum(ev_c["sum"])
    assert abs(contiguous["mpjpe_optimised_mm"] - cyclic["mpjpe_optimised_mm"]) < 0.5
