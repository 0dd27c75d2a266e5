"""ctypes loader for oracle/build/libukf_oracle.so -- the CPU ORACLE (test infrastructure only).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
PARITY UNPINNED (see oracle/ukf_oracle.hpp).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "build", "libukf_oracle.so")
_lib = None


class Config(C.Structure):
    _fields_ = [("mean_tol", C.c_double), ("mean_max_it", C.c_int32), ("gate_chi2", C.c_double),
                ("min_dt", C.c_double), ("max_dt", C.c_double)]


def build(force: bool = False) -> str:
    """Compile the oracle with the committed Makefile (g++, no external dependency)."""
    src = [os.path.join(_HERE, f) for f in ("ukf_oracle_capi.cpp", "ukf_oracle.hpp", "Makefile")]
    stale = (not os.path.exists(_LIB_PATH)) or any(os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in src)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.ukfo_earthw.restype = C.c_double
        _lib.ukfo_max_threads.restype = C.c_int
    return _lib


def native_lib():
    """-O3 -march=native build of the same oracle for bench.py's cpu_baseline leg, compiled on THIS host (the
    GPU box's CPU differs from the build container's, so the library is never shipped).  None if g++ fails."""
    import tempfile
    out = os.path.join(tempfile.gettempdir(), f"ukf_oracle_native_{os.getuid()}", "libukf_oracle_native.so")
    try:
        subprocess.check_call(["make", "-s", "-B", "-C", _HERE, "native", f"NATIVE_OUT={out}"])
        h = C.CDLL(out)
        h.ukfo_earthw.restype = C.c_double
        h.ukfo_max_threads.restype = C.c_int
        return h
    except Exception:
        return None


class using:
    """Context manager: route the wrappers of this module through another build of the oracle library."""

    def __init__(self, handle):
        self.handle = handle

    def __enter__(self):
        global _lib
        lib()
        self.prev, _lib = _lib, self.handle
        return self

    def __exit__(self, *exc):
        global _lib
        _lib = self.prev
        return False


def default_config(**over) -> Config:
    c = Config()
    lib().ukfo_default_config(C.byref(c))
    for k, v in over.items():
        setattr(c, k, v)
    return c


def _d(a):
    return a.ctypes.data_as(C.POINTER(C.c_double)) if a is not None else None


def _f64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None:
        a = a.reshape(shape)
    return a


def max_threads() -> int:
    return int(lib().ukfo_max_threads())


def pose_predict(mu, cov, R, acc_mu, acc_cov, dt, prec=0, cfg=None, threads=1):
    """In-place on copies; returns (mu, cov, status). R [12,12] or [n,12,12]; acc_mu None or [n,3];
    acc_cov [3,3] or [n,3,3]; dt scalar or [n]."""
    mu = _f64(mu).copy(); cov = _f64(cov).copy()
    n = mu.shape[0]
    R = _f64(R); acc_cov = _f64(acc_cov if acc_cov is not None else np.eye(3))
    dt = _f64(np.atleast_1d(dt))
    am = _f64(acc_mu) if acc_mu is not None else None
    st = np.zeros(n, dtype=np.uint32)
    cfg = cfg or default_config()
    lib().ukfo_pose_predict(C.c_int64(n), C.c_int(prec), _d(mu), _d(cov), _d(R), C.c_int(R.ndim == 3),
                            _d(am), _d(acc_cov), C.c_int(acc_cov.ndim == 3), _d(dt), C.c_int(dt.size > 1),
                            C.byref(cfg), st.ctypes.data_as(C.POINTER(C.c_uint32)), C.c_int(threads))
    return mu, cov, st


def pose_update(mu, cov, model, z, Q, prec=0, cfg=None, threads=1):
    """model: int or int32[n] (negative = inactive); z [n,3]; Q [n,3,3]."""
    mu = _f64(mu).copy(); cov = _f64(cov).copy()
    n = mu.shape[0]
    model = np.ascontiguousarray(np.atleast_1d(model), dtype=np.int32)
    z = _f64(z); Q = _f64(Q)
    st = np.zeros(n, dtype=np.uint32)
    cfg = cfg or default_config()
    lib().ukfo_pose_update(C.c_int64(n), C.c_int(prec), _d(mu), _d(cov),
                           model.ctypes.data_as(C.POINTER(C.c_int32)), C.c_int(model.size > 1), _d(z), _d(Q),
                           C.byref(cfg), st.ctypes.data_as(C.POINTER(C.c_uint32)), C.c_int(threads))
    return mu, cov, st


def orient_predict(mu, cov, R, acc, gyro, tau_g, tau_a, earth, dt, prec=0, cfg=None, threads=1):
    mu = _f64(mu).copy(); cov = _f64(cov).copy()
    n = mu.shape[0]
    R = _f64(R); acc = _f64(acc); gyro = _f64(gyro); earth = _f64(earth)
    dt = _f64(np.atleast_1d(dt))
    st = np.zeros(n, dtype=np.uint32)
    cfg = cfg or default_config()
    lib().ukfo_orient_predict(C.c_int64(n), C.c_int(prec), _d(mu), _d(cov), _d(R), C.c_int(R.ndim == 3), _d(acc),
                              _d(gyro), C.c_double(tau_g), C.c_double(tau_a), _d(earth), _d(dt),
                              C.c_int(dt.size > 1), C.byref(cfg), st.ctypes.data_as(C.POINTER(C.c_uint32)),
                              C.c_int(threads))
    return mu, cov, st


def orient_update(mu, cov, z, Q, active=None, prec=0, cfg=None, threads=1):
    mu = _f64(mu).copy(); cov = _f64(cov).copy()
    n = mu.shape[0]
    z = _f64(z); Q = _f64(Q)
    act = np.ascontiguousarray(active, dtype=np.uint8) if active is not None else None
    st = np.zeros(n, dtype=np.uint32)
    cfg = cfg or default_config()
    lib().ukfo_orient_update(C.c_int64(n), C.c_int(prec), _d(mu), _d(cov),
                             act.ctypes.data_as(C.POINTER(C.c_uint8)) if act is not None else None, _d(z), _d(Q),
                             C.byref(cfg), st.ctypes.data_as(C.POINTER(C.c_uint32)), C.c_int(threads))
    return mu, cov, st


def gate_timestamps(ts_us, last_us, min_dt=1e-9, max_dt=np.finfo(np.float64).max):
    ts = np.ascontiguousarray(ts_us, dtype=np.int64)
    last = np.ascontiguousarray(last_us, dtype=np.int64).copy()
    n = ts.size
    dt = np.zeros(n); st = np.zeros(n, dtype=np.uint32)
    lib().ukfo_gate_timestamps(C.c_int64(n), ts.ctypes.data_as(C.POINTER(C.c_int64)),
                               last.ctypes.data_as(C.POINTER(C.c_int64)), C.c_double(min_dt), C.c_double(max_dt),
                               _d(dt), st.ctypes.data_as(C.POINTER(C.c_uint32)))
    return last, dt, st


def orient_rotation_rate(mu, gyro, earth):
    mu = _f64(mu); gyro = _f64(gyro); earth = _f64(earth)
    out = np.zeros((mu.shape[0], 3))
    lib().ukfo_orient_rotation_rate(C.c_int64(mu.shape[0]), _d(mu), _d(gyro), _d(earth), _d(out))
    return out


# unit-level hooks ----------------------------------------------------------------
def so3_exp(v, scale=1.0, f32=False):
    v = _f64(v); q = np.zeros(4)
    (lib().ukfo_so3_exp_f32 if f32 else lib().ukfo_so3_exp)(_d(v), C.c_double(scale), _d(q))
    return q


def so3_log(q):
    q = _f64(q); v = np.zeros(3)
    lib().ukfo_so3_log(_d(q), _d(v))
    return v


def quat_rotate(q, v):
    q = _f64(q); v = _f64(v); r = np.zeros(3)
    lib().ukfo_quat_rotate(_d(q), _d(v), _d(r))
    return r


def quat_to_matrix(q):
    q = _f64(q); R = np.zeros((3, 3))
    lib().ukfo_quat_to_matrix(_d(q), _d(R))
    return R


def pose_boxplus(x, d):
    x = _f64(x).copy(); d = _f64(d)
    lib().ukfo_pose_boxplus(_d(x), _d(d))
    return x


def pose_boxminus(x, y):
    x = _f64(x); y = _f64(y); d = np.zeros(12)
    lib().ukfo_pose_boxminus(_d(x), _d(y), _d(d))
    return d


def orient_boxplus(x, d):
    x = _f64(x).copy(); d = _f64(d)
    lib().ukfo_orient_boxplus(_d(x), _d(d))
    return x


def orient_boxminus(x, y):
    x = _f64(x); y = _f64(y); d = np.zeros(13)
    lib().ukfo_orient_boxminus(_d(x), _d(y), _d(d))
    return d


def cholesky12(A):
    A = _f64(A); L = np.zeros((12, 12))
    rc = lib().ukfo_cholesky12(_d(A), _d(L))
    return L, rc == 0


def pose_process(x, acc, dt):
    x = _f64(x).copy()
    a = _f64(acc) if acc is not None else None
    lib().ukfo_pose_process(_d(x), _d(a), C.c_double(dt))
    return x


def orient_process(x, acc, gyro, tau_g, tau_a, earth, dt):
    x = _f64(x).copy()
    lib().ukfo_orient_process(_d(x), _d(_f64(acc)), _d(_f64(gyro)), C.c_double(tau_g), C.c_double(tau_a),
                              _d(_f64(earth)), C.c_double(dt))
    return x


def earthw() -> float:
    return float(lib().ukfo_earthw())
