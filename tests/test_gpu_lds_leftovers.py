"""No kernel of the engine may depend on LDS it did not write itself.

tests/cpp/lds_poison.hip fills the LDS of every CU with a bit pattern; the engine then runs the launches the BASELINE
configurations use (fused cycle, separate prediction and update, multi-cycle, per-filter models, event rounds = the
indirect kernel) and must give the SAME BITS
after a NaN pattern, an Inf pattern and zeros.  Found in round 3 by tests/fuzz_parity.py as a box-dependent Cholesky failure:
an alignment pad between two LDS regions was read as "leftover times the table's exact-zero row" -- NaN whenever the leftover
of an earlier kernel happened to be NaN or Inf (Layout16::LAF_PAD in ukf_kernel16.hpp)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "tests", "cpp", "build", "liblds_poison.so")
PATTERNS = (0x7FC00000, 0x7F800000, 0xFFFFFFFF, 0x00000000)     # float NaN, float +Inf, a NaN in both precisions, zero


def _poison():
    if not os.path.exists(LIB):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "cpp")])
    lib = C.CDLL(LIB)
    lib.lds_poison.argtypes = [C.c_uint32]
    return lib.lds_poison


def _run(spe, model, prec, G, n=4099):
    import torch
    s = spe.synth
    tdt = torch.float64 if prec == 0 else torch.float32
    dev = lambda x: torch.from_numpy(np.ascontiguousarray(x.reshape(x.shape[0], -1))).to("cuda", tdt)   # noqa: E731
    out = []
    if model == "pose":
        mu, cov = s.pose_initial(n)
        e = spe.BatchPoseUKF(n, precision=prec, lanes_per_filter=G)
        e.initialize(mu, cov)
        acc, z, Q = s.pose_cycle_inputs(n, 0, mu[:, :3], random_q=True)
        e.set_acceleration(acc, 0.01 * np.eye(3))
        e.cycle(0.01, spe.MEAS_POS3, z, Q)
        models = s.pose_mixed_models(n, 1)
        zz = s.pose_measurement_for_model(mu, models, z - mu[:, :3])
        e.update(models, zz, Q)                                  # per-filter models incl. the SO(3) sigma-point path
        e.set_acceleration(np.full((n, 3), np.nan), None)        # constant-velocity branch: rotated noise, shaped-noise table
        e.predict(0.02)
        z_t, Q_t = dev(np.stack([z, z])), dev(np.stack([Q.reshape(n, 9)] * 2))
        torch.cuda.synchronize()
        e.cycle_multi_dev(2, 0.01, spe.MEAS_VEL3, z_t.reshape(2, n, 3), Q_t.reshape(2, n, 9), 2, 0)
    else:
        mu, cov = s.orient_initial(n)
        e = spe.BatchOrientationUKF(n, s.ORIENT_TAU, s.ORIENT_TAU, s.ORIENT_LATITUDE, precision=prec, lanes_per_filter=G)
        R = s.orient_process_noise()
        e.set_process_noise(R)
        e.initialize(mu, cov)
        gyro, acc, z, Q = s.orient_cycle_inputs(n, 0, mu[:, :4])
        e.set_orient_inputs(gyro, acc)
        e.cycle(0.01, spe.MEAS_ORIENT_BODYVEL3, z, Q)
        R2 = R.copy(); R2[0, 0] *= 3.0; R2[4, 4] *= 2.0            # anisotropic blocks: the rotated-noise path
        e.set_process_noise(R2)
        e.predict(0.02)
        e.update(spe.MEAS_ORIENT_BODYVEL3, z, Q)
        z_t, Q_t = dev(np.stack([z, z])), dev(np.stack([Q.reshape(n, 9)] * 2))
        torch.cuda.synchronize()
        e.cycle_multi_dev(2, 0.01, spe.MEAS_ORIENT_BODYVEL3, z_t.reshape(2, n, 3), Q_t.reshape(2, n, 9), 2, 0)
    m, c, _ = e.state()
    st = e.status()
    # the indirect instantiation (event rounds): two samples for every third filter, a third one for some, any arrival order
    rng = np.random.default_rng(5)
    f_ev = np.concatenate([np.arange(0, n, 3), np.arange(0, n, 3), np.arange(0, n, 7)])
    t_ev = np.concatenate([np.full(len(range(0, n, 3)), 1_010_000), np.full(len(range(0, n, 3)), 1_020_000), np.full(len(range(0, n, 7)), 1_030_000)])
    perm = rng.permutation(f_ev.size)
    f_ev, t_ev = f_ev[perm], t_ev[perm]
    model_ev = spe.MEAS_POS3 if model == "pose" else spe.MEAS_ORIENT_BODYVEL3
    e.process_events(f_ev, t_ev, np.full(f_ev.size, model_ev, dtype=np.int32), z[f_ev], Q[f_ev])
    m2, c2, _ = e.state()
    st2 = e.status()
    e.close()
    return m, c, st, m2, c2, st2


@pytest.mark.parametrize("G", [16, 64])      # the tuned layout and the one-wavefront-per-filter ablation (fp32 in the shipped library)
@pytest.mark.parametrize("prec", [0, 1])
@pytest.mark.parametrize("model", ["pose", "orient"])
def test_results_do_not_depend_on_lds_leftovers(spe, model, prec, G):
    poison = _poison()
    ref = None
    for pat in PATTERNS:
        assert poison(pat) == 0
        got = _run(spe, model, prec, G)
        m, c, st, m2, c2, _ = got
        assert np.isfinite(m).all() and np.isfinite(c).all() and (st == 0).all(), (model, prec, hex(pat))
        assert np.isfinite(m2).all() and np.isfinite(c2).all() and not np.array_equal(m, m2), (model, prec, hex(pat))
        if ref is None:
            ref = got
        else:
            assert all(np.array_equal(x, y) for x, y in zip(got, ref)), (model, prec, hex(pat))
