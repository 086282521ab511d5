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


sys.path.insert(0, GOLDEN)
import compact  # noqa: E402  (tests/golden/compact.py: which arrays are stored whole, which as samples, the shared DeltaBlock weights)


def load_golden(name):
    """name -> tensor, or a `compact.Sampled` comparison target (assert_close / err_stats compare its stored positions)."""
    return compact.load(os.path.join(GOLDEN, name))


def assert_close(got, want, rtol=RTOL, atol=ATOL, what=""):
    if isinstance(want, compact.Sampled):     # a reference tensor of which one element in 16 is stored: compare those positions
        got, want, what = want.take(got.detach().float().cpu()), want.values, what + " [1/16 sample]"
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
