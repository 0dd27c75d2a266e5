"""slam-pose_estimation_amd -- MI355X-native batched UKF engine for the pose_estimation hot path.

Holds only what the path needs: csrc/ (HIP kernels + the C-ABI of include/ukf_batch.h), the ctypes
binding (engine.py), the build recipe (build.py) and the deterministic synthetic inputs (synth.py).
The directory name carries a hyphen, so import it through the repo-root shim:

    import slam_pose_estimation_amd as spe
"""
from . import synth  # noqa: F401
from .sharding import gather_means, shard_range  # noqa: F401
from .build import build_engine  # noqa: F401
from .engine import *  # noqa: F401,F403
from .engine import BatchOrientationUKF, BatchPoseUKF, BatchUKF, Config, UKFGroup, UkfbError, layout_supported, load_library  # noqa: F401
