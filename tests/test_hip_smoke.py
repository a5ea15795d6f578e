"""The driver's round-end smoke call, kept green by the suite: `__graft_entry__.smoke()` runs one small batch through the HIP path on
cuda:0 and checks it against the oracle (one evaluation to rounding; a whole L-BFGS stage on its closure values, counts, reached loss)."""
import pytest

pytestmark = pytest.mark.gpu


def test_graft_entry_smoke(capsys):
    import torch
    if not torch.cuda.is_available():
        pytest.fail("gpu tests need a HIP device")
    import __graft_entry__ as entry
    entry.smoke()
    assert "smoke ok" in capsys.readouterr().out
