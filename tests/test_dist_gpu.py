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
