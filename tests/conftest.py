"""Test configuration: paths, the `gpu` marker, shared fixtures."""

import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
for p in (ROOT / 'torch-darktable_amd', ROOT / 'oracle', ROOT):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def oracle():
    import tdk_oracle

    tdk_oracle.build()
    return tdk_oracle


@pytest.fixture(scope='session')
def td():
    """The product package; importing it loads libtdk_hip.so (built if missing)."""
    import __graft_entry__  # noqa: F401  (build on first use)

    __graft_entry__.ensure_built()
    import torch_darktable

    return torch_darktable


@pytest.fixture(scope='session')
def scene():
    import torch
    from torch_darktable.synthetic import synthetic_rgb

    def make(h, w, seed=1234, noise=0.02):
        torch.manual_seed(seed)
        return synthetic_rgb(h, w, seed, 'cpu', noise).numpy()

    return make
