"""GPU: whole training steps through the host mirror (reference API) vs the oracle."""
import numpy as np
import pytest

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu


def test_front_dpcl_training_step_matches_oracle():
    from tests.smoke_step import run_smoke
    errs = run_smoke(torch, np, verbose=False)
    assert max(errs.values()) < 2e-4, errs
