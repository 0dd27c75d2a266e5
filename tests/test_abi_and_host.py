"""CPU-side checks of the drop-in boundary: the shared library loads, exports every symbol that
include/ukf_batch.h declares (no compute call is made without a GPU), and fails loudly -- never
falls back to a CPU path -- when no HIP device is present."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "ukf_batch.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ukfb_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_list_the_same_entry_points(spe):
    assert header_symbols() == sorted(spe.engine.EXPORTS)


def test_library_exports_every_declared_symbol(spe):
    lib = spe.load_library()
    missing = [s for s in header_symbols() if not hasattr(lib, s)]
    assert missing == []


def test_header_is_plain_c_and_links_from_c(spe, tmp_path):
    """The boundary is a C ABI: the header must compile as strict C99 and a C program must link the library
    (without a GPU ukfb_create reports an error code instead of crashing)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "c_abi.c"
    src.write_text('#include <stdio.h>\n#include "ukf_batch.h"\n'
                   'int main(void) { ukfb_engine* e = 0;\n'
                   '  int rc = ukfb_create(&e, UKFB_MODEL_POSE, UKFB_F64, 4, 0, 0);\n'
                   '  printf("%d %s\\n", rc, rc == UKFB_OK ? "ok" : ukfb_last_error());\n'
                   '  if (rc == UKFB_OK) ukfb_destroy(e);\n  return 0; }\n')
    exe = tmp_path / "c_abi"
    lib = os.path.join(root, "slam-pose_estimation_amd", "lib")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(root, "include"),
                           str(src), "-o", str(exe), "-L", lib, "-lukf_batch", "-Wl,-rpath," + lib])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.strip() != ""


def test_default_config_matches_reference_defaults(spe):
    import ctypes as C
    lib = spe.load_library()
    c = spe.Config()
    assert lib.ukfb_default_config(C.byref(c)) == 0
    assert c.mean_tol == 1e-6                      # ukfom meanSigmaPoints
    assert c.min_time_delta == 1.0e-9              # UnscentedKalmanFilter.hpp:31
    assert c.max_time_delta == np.finfo(np.float64).max   # UnscentedKalmanFilter.hpp:32
    assert c.gate_chi2 < 0                         # accept_any_mahalanobis_distance
    assert c.lanes_per_filter == 16
    assert c.wide_arithmetic == 0 and c.full_update_check == 0 and c.split_streams == 1 and c.bucket_models == 1
    # ukfom's iteration cap, and the CPU oracle's (oracle/ukf_oracle.hpp:78): the engine must not give up earlier
    from oracle import capi
    assert c.mean_max_iter == 10000 == capi.default_config().mean_max_it
    # layouts of this build: the tuned one in both precisions, the one-wavefront-per-filter ablation in fp32
    assert lib.ukfb_layout_supported(spe.F64, 16) == 1 and lib.ukfb_layout_supported(spe.F32, 0) == 1
    assert lib.ukfb_layout_supported(spe.F32, 32) == 1 and lib.ukfb_layout_supported(spe.F32, 64) == 1
    assert lib.ukfb_layout_supported(spe.F64, 48) == 0 and lib.ukfb_layout_supported(7, 16) == 0


def test_status_bits_agree_between_engine_binding_and_oracle(spe, onp):
    for name in ("ST_SKIPPED_FIRST_TS", "ST_SKIPPED_SMALL_DT", "ST_ERR_NEG_DT", "ST_ERR_DT_TOO_LARGE",
                 "ST_ERR_NONFINITE_MEAS", "ST_ERR_CHOLESKY", "ST_WARN_MEAN_NOCONV", "ST_UNINITIALISED", "ST_INACTIVE",
                 "ST_REJECTED_GATE"):
        assert getattr(spe, name) == getattr(onp, name)
    text = open(os.path.join(ROOT, "include", "ukf_batch.h")).read()
    for name, bit in re.findall(r"UKFB_(ST_[A-Z_]+) = 1u << (\d+)", text):
        assert getattr(spe, name) == 1 << int(bit)
    for name, val in re.findall(r"UKFB_(MEAS_[A-Z0-9_]+) = (-?\d+)", text):
        assert getattr(spe, name) == int(val)


def test_engine_refuses_to_run_without_a_gpu(spe):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(spe.UkfbError) as ei:
        spe.BatchPoseUKF(8)
    assert "no usable HIP device" in str(ei.value)


def test_missing_library_is_a_loud_error(spe, monkeypatch, tmp_path):
    monkeypatch.setattr(spe.engine, "LIB_PATH", str(tmp_path / "nope.so"))
    monkeypatch.setattr(spe.engine, "_lib", None)
    with pytest.raises(spe.UkfbError):
        spe.load_library()


def test_shard_ranges_partition_the_batch(spe):
    for total in (0, 1, 7, 1_048_576, 1_000_003):
        for world in (1, 2, 3, 8):
            nxt = 0
            for r in range(world):
                first, count = spe.shard_range(total, world, r)
                assert first == nxt and count >= 0
                nxt = first + count
            assert nxt == total
    assert spe.shard_range(1_048_576, 8, 3) == (393216, 131072)


def test_synthetic_inputs_are_counter_based(spe):
    """Any shard regenerates exactly the slice of the full batch (SplitMix64 keyed by filter id)."""
    mu, cov = spe.synth.pose_initial(64)
    mu2, cov2 = spe.synth.pose_initial(16, first=24)
    assert np.array_equal(mu[24:40], mu2) and np.array_equal(cov[24:40], cov2)
    assert np.abs(np.linalg.norm(mu[:, 3:7], axis=1) - 1).max() < 1e-15
    assert (np.linalg.eigvalsh(cov) > 0).all() and np.array_equal(cov, np.swapaxes(cov, 1, 2))
    a1 = spe.synth.pose_cycle_inputs(8, 5, first=100)[0]
    a2 = spe.synth.pose_cycle_inputs(32, 5, first=92)[0][8:16]
    assert np.array_equal(a1, a2)


@pytest.mark.parametrize("std", ["c++11", "c++17"])
def test_caller_text_written_against_the_reference_api_compiles(tmp_path, std):
    """north_star: 'drops into Rock as-is'.  tests/cpp/reference_caller_text.cpp holds caller statements typed against
    the reference's Eigen / MTK API (assignable blocks, setZero, MTK::SO3<double>::exp, MTK::subblock / setDiagonal,
    PoseUKF::MTK_UKF::cov, WState operator+ / operator-, boxplus / boxminus, Eigen::Matrix<...> spellings);
    they must compile -- warnings as errors -- against include/ unchanged and give the reference's results."""
    import subprocess
    exe = tmp_path / "caller_text"
    subprocess.check_call(["g++", f"-std={std}", "-O1", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "reference_caller_text.cpp"), "-o", str(exe),
                           "-L", os.path.join(ROOT, "slam-pose_estimation_amd", "lib"), "-lukf_batch",
                           "-Wl,-rpath," + os.path.join(ROOT, "slam-pose_estimation_amd", "lib")])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0 and "ok" in out.stdout, out.stdout + out.stderr
    # the MTK::SO3 host conveniences (pose_estimation/Manifold.hpp: exp, log, boxplus with scale, boxminus) against scipy's
    # Rotation -- an independent implementation (MTK itself is not vendored with the reference: unpinned)
    from scipy.spatial.transform import Rotation
    dump = subprocess.run([str(exe), "dump"], capture_output=True, text=True, timeout=60)
    assert dump.returncode == 0
    rows = [[float(t) for t in line.split() if t[0] in "-0123456789" and t not in ("->",)] for line in dump.stdout.strip().splitlines()]
    kinds = [line.split()[0] for line in dump.stdout.strip().splitlines()]
    assert kinds.count("exp") == 8 and kinds.count("boxplus") == 8
    prev_q = None
    for kind, r in zip(kinds, rows):
        if kind == "exp":
            v, q, back = np.array(r[0:3]), np.array(r[3:7]), np.array(r[7:10])
            q_ref = Rotation.from_rotvec(v).as_quat()          # (x, y, z, w), w >= 0 for |v| <= pi
            assert np.abs(q - q_ref).max() < 1e-15
            # MTK's log is atan based: angle in (-pi, pi), i.e. scipy's rotation vector for |v| < pi
            assert np.abs(back - Rotation.from_quat(q_ref).as_rotvec()).max() < 1e-14
            prev_q = q
        else:
            w, scale, p, diff = np.array(r[0:3]), r[3], np.array(r[4:8]), np.array(r[8:11])
            p_ref = (Rotation.from_quat(prev_q) * Rotation.from_rotvec(scale * w)).as_quat()
            p_ref = p_ref if np.dot(p_ref, p) > 0 else -p_ref
            assert np.abs(p - p_ref).max() < 1e-15
            # boxminus = log(q^-1 p) = the scaled increment (|scale w| < pi here)
            assert np.abs(diff - scale * w).max() < 1e-14


def test_new_entry_points_reject_a_null_engine(spe):
    """the multi-cycle and uniform-Q entry points return an error code for a NULL handle (no GPU needed, nothing is launched)"""
    import ctypes as C
    lib = spe.load_library()
    null = C.c_void_p(None)
    d9 = (C.c_double * 9)()
    i1 = (C.c_int32 * 1)()
    assert lib.ukfb_cycle_multi_dev(null, 1, C.c_double(0.01), 0, 1, 0, null, null, null, null) != 0
    assert lib.ukfb_cycle_multi(null, 1, C.c_double(0.01), 0, null, null, d9, d9) != 0
    assert lib.ukfb_cycle_multi_mixed_dev(null, 1, C.c_double(0.01), 1, 0, null, null, null, null, null) != 0
    assert lib.ukfb_cycle_schedule_dev(null, 1, d9, i1, 1, 0, null, null, null, null) != 0
    assert lib.ukfb_cycle_uniform_q(null, C.c_double(0.01), 0, d9, d9) != 0
    assert lib.ukfb_cycle_uniform_q_dev(null, C.c_double(0.01), 0, null, null) != 0
    assert lib.ukfb_update_uniform_q(null, 0, d9, d9, null) != 0


def test_group_entry_points_without_a_gpu(spe):
    """Device groups (ukfb_group_*): the shard split is the one of sharding.py / bench.py, NULL handles are rejected, and
    without a HIP device creation fails loudly (no CPU path) -- nothing is launched."""
    import ctypes as C
    import torch
    lib = spe.load_library()
    for total in (8, 1_048_576, 1_000_003):
        for world in (1, 2, 3, 8):
            for r in range(world):
                f, c = C.c_int64(-1), C.c_int64(-1)
                assert lib.ukfb_group_shard_range(C.c_int64(total), world, r, C.byref(f), C.byref(c)) == 0
                assert (f.value, c.value) == spe.shard_range(total, world, r)
    assert lib.ukfb_group_shard_range(C.c_int64(8), 2, 2, None, None) != 0
    null = C.c_void_p(None)
    assert lib.ukfb_group_size(null) == -1
    assert lib.ukfb_group_sync(null) != 0 and lib.ukfb_group_cycle_dev(null, C.c_double(0.01), 0, null, null) != 0
    assert lib.ukfb_group_gather_means(null, null) != 0 and lib.ukfb_group_destroy(null) == 0
    assert lib.ukfb_group_cycle_mixed_dev(null, C.c_double(0.01), null, null, null) != 0
    assert lib.ukfb_group_cycle_timestamps(null, null, null, null, null) != 0
    assert lib.ukfb_group_process_events(null, C.c_int64(0), null, null, null, null, null, null, null) != 0
    if not torch.cuda.is_available():
        with pytest.raises(spe.UkfbError) as ei:
            spe.UKFGroup(spe.MODEL_POSE, spe.F64, 64, [0, 0])
        assert "no usable HIP device" in str(ei.value)
