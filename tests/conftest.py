"""pytest configuration: markers + shared fixtures."""
import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


@pytest.fixture(scope="session")
def golden():
    return np.load(os.path.join(GOLDEN, "unet_g12_b2.npz"))


@pytest.fixture(scope="session")
def golden_b():
    """Second reference fixture (recipe G1.2-B: gammas of both signs, small BatchNorm variances, audio x4, B=3)."""
    return np.load(os.path.join(GOLDEN, "unet_g12b_b3.npz"))


@pytest.fixture(scope="session")
def recipe_sd_b():
    from calipsync_amd import recipe
    return recipe.make_state_dict_b()


@pytest.fixture(scope="session")
def recipe_sd():
    from calipsync_amd import recipe
    return recipe.make_state_dict()


def sample_indices(numel: int, n: int = 4096) -> np.ndarray:
    """Same strided sample positions tests/golden/make_golden.py used."""
    return (np.arange(n, dtype=np.int64) * 2654435761) % numel
