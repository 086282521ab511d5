import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# north-star tolerance (BASELINE.json): rtol=1e-3 / atol=1e-4 in fp32
RTOL, ATOL = 1e-3, 1e-4


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name))
    return {k: torch.from_numpy(z[k]) for k in z.files}


def assert_close(got, want, rtol=RTOL, atol=ATOL, what=""):
    got, want = got.detach().float().cpu(), want.detach().float().cpu()
    assert got.shape == want.shape, f"{what}: shape {tuple(got.shape)} vs {tuple(want.shape)}"
    err = (got - want).abs()
    bad = err > (atol + rtol * want.abs())
    assert not bad.any(), (f"{what}: {int(bad.sum())}/{bad.numel()} outside rtol={rtol} atol={atol}; "
                           f"max abs err {float(err.max()):.3e}, mean {float(err.mean()):.3e}, "
                           f"ref absmax {float(want.abs().max()):.3e}")


@pytest.fixture(scope="session")
def golden_small():
    return load_golden("ddpm_small.npz")


@pytest.fixture(scope="session")
def golden_celeba():
    return load_golden("ddpm_celeba.npz")
