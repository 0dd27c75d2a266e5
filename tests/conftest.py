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


@pytest.fixture(autouse=True)
def _skip_unbuilt_layouts(request):
    """The one-wavefront-per-filter layouts (G = 32 / 64) ship in fp32 only; their fp64 instantiations are a diagnostic
    build option (make GENERIC_F64=1, include/ukf_batch.h).  Parametrised (prec, G) cases the library cannot run skip."""
    cs = getattr(request.node, "callspec", None)
    if cs is None or "G" not in cs.params or request.node.get_closest_marker("gpu") is None:
        return
    import slam_pose_estimation_amd as m
    prec = cs.params.get("prec", 0)
    if not m.layout_supported(int(prec), int(cs.params["G"])):
        pytest.skip(f"lanes_per_filter={cs.params['G']} at precision {prec} is not in this build (GENERIC_F64=0)")


def max_abs(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b)))) if np.size(a) else 0.0
