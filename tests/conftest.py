import torch  # noqa: F401  (must load before the engine library: one HIP runtime per process)
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def spe():
    import slam_pose_estimation_amd as m
    return m


@pytest.fixture(scope="session")
def oracle():
    from oracle import capi
    capi.lib()
    return capi


@pytest.fixture(scope="session")
def onp():
    from oracle import ukf_numpy
    return ukf_numpy


def max_abs(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b)))) if np.size(a) else 0.0
