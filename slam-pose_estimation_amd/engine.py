"""ctypes binding of the C-ABI in include/ukf_batch.h (lib/libukf_batch.so).

This is plumbing: every numeric operation happens in the HIP kernels behind the C-ABI.  There is no
CPU fallback -- if the shared library or a HIP device is missing, construction raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("UKFB_LIB", os.path.join(_HERE, "lib", "libukf_batch.so"))  # UKFB_LIB: debug builds only

MODEL_POSE, MODEL_ORIENT = 0, 1
F64, F32 = 0, 1

MEAS_NONE = -1
MEAS_POS3, MEAS_POS_XY, MEAS_POS_Z, MEAS_ORIENT_SO3, MEAS_VEL3 = 0, 1, 2, 3, 4
MEAS_VEL_XY, MEAS_VEL_Z, MEAS_XVEL_YAWVEL, MEAS_ANGVEL3, MEAS_ORIENT_BODYVEL3 = 5, 6, 7, 8, 9

ST_OK = 0
ST_SKIPPED_FIRST_TS = 1 << 0
ST_SKIPPED_SMALL_DT = 1 << 1
ST_ERR_NEG_DT = 1 << 2
ST_ERR_DT_TOO_LARGE = 1 << 3
ST_ERR_NONFINITE_MEAS = 1 << 4
ST_ERR_CHOLESKY = 1 << 5
ST_WARN_MEAN_NOCONV = 1 << 6
ST_UNINITIALISED = 1 << 7
ST_INACTIVE = 1 << 8
ST_REJECTED_GATE = 1 << 9

# every symbol include/ukf_batch.h declares (tests check the library exports all of them)
EXPORTS = [
    "ukfb_default_config", "ukfb_layout_supported", "ukfb_create", "ukfb_create_on_stream", "ukfb_destroy", "ukfb_last_error", "ukfb_set_config", "ukfb_get_config",
    "ukfb_sync", "ukfb_describe", "ukfb_initialize", "ukfb_get_state", "ukfb_get_status", "ukfb_get_status_summary",
    "ukfb_set_last_measurement_time", "ukfb_get_last_measurement_time", "ukfb_device_views",
    "ukfb_set_process_noise", "ukfb_set_process_noise_per_filter", "ukfb_get_process_noise",
    "ukfb_pose_set_acceleration", "ukfb_pose_bind_acceleration_dev", "ukfb_orient_set_params",
    "ukfb_orient_set_inputs", "ukfb_orient_bind_inputs_dev", "ukfb_orient_get_rotation_rate", "ukfb_predict",
    "ukfb_predict_dt", "ukfb_predict_timestamps", "ukfb_predict_dt_dev", "ukfb_predict_timestamps_dev",
    "ukfb_update", "ukfb_update_mixed", "ukfb_update_dev", "ukfb_cycle", "ukfb_cycle_dev", "ukfb_cycle_multi_dev", "ukfb_cycle_multi",
    "ukfb_cycle_schedule_dev", "ukfb_cycle_multi_mixed_dev", "ukfb_update_uniform_q", "ukfb_cycle_uniform_q",
    "ukfb_cycle_uniform_q_dev",
    "ukfb_last_launch_info",
    "ukfb_last_model_groups",
    "ukfb_timer_begin", "ukfb_timer_end", "ukfb_pose_export_body_states", "ukfb_pose_import_body_states",
    "ukfb_cycle_timestamps", "ukfb_cycle_timestamps_dev", "ukfb_process_events", "ukfb_process_events_dev",
    # device groups (one process, several GPUs)
    "ukfb_group_shard_range", "ukfb_group_create", "ukfb_group_destroy", "ukfb_group_size", "ukfb_group_shard",
    "ukfb_group_set_config", "ukfb_group_initialize", "ukfb_group_get_state", "ukfb_group_get_status",
    "ukfb_group_get_status_summary", "ukfb_group_set_process_noise", "ukfb_group_pose_set_acceleration",
    "ukfb_group_orient_set_params", "ukfb_group_orient_set_inputs", "ukfb_group_predict", "ukfb_group_update",
    "ukfb_group_cycle", "ukfb_group_pose_bind_acceleration_dev", "ukfb_group_orient_bind_inputs_dev",
    "ukfb_group_cycle_dev", "ukfb_group_cycle_multi_dev", "ukfb_group_cycle_mixed_dev", "ukfb_group_cycle_timestamps",
    "ukfb_group_process_events", "ukfb_group_sync", "ukfb_group_timer_begin",
    "ukfb_group_timer_end", "ukfb_group_gather_means", "ukfb_group_last_gather_exchange",
]
BODY_STATE_SCALARS = 49


class Config(C.Structure):
    _fields_ = [("mean_tol", C.c_double), ("mean_max_iter", C.c_int32), ("gate_chi2", C.c_double),
                ("min_time_delta", C.c_double), ("max_time_delta", C.c_double), ("lanes_per_filter", C.c_int32),
                ("bucket_models", C.c_int32), ("split_streams", C.c_int32), ("wide_arithmetic", C.c_int32),
                ("full_update_check", C.c_int32)]


class UkfbError(RuntimeError):
    pass


_lib = None


def load_library():
    """Load libukf_batch.so; raises if it has not been built (no fallback path exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise UkfbError(f"{LIB_PATH} is missing: build it with __graft_entry__.build() "
                            f"(make -C slam-pose_estimation_amd/csrc); there is no CPU fallback")
        lib = C.CDLL(LIB_PATH)
        lib.ukfb_last_error.restype = C.c_char_p
        _lib = lib
    return _lib


def _chk(rc: int, what: str):
    if rc != 0:
        msg = load_library().ukfb_last_error()
        raise UkfbError(f"{what} failed with code {rc}: {msg.decode() if msg else ''}")


def _pd(a):
    return a.ctypes.data_as(C.POINTER(C.c_double)) if a is not None else None


def _f64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None:
        a = a.reshape(shape)
    return a


def _devptr(x):
    """Accept an int address, a torch tensor (data_ptr) or None."""
    if x is None:
        return None
    if hasattr(x, "data_ptr"):
        return C.c_void_p(x.data_ptr())
    return C.c_void_p(int(x))


def _torch_current_stream(device: int):
    """hipStream_t of torch's current stream on `device` (0 = the default stream), or None when torch is not in use."""
    import sys
    torch = sys.modules.get("torch")
    if torch is None:
        return None
    try:
        if not torch.cuda.is_available():
            return None
        return int(torch.cuda.current_stream(device).cuda_stream)
    except Exception:
        return None


def layout_supported(precision: int, lanes_per_filter: int) -> bool:
    return bool(load_library().ukfb_layout_supported(C.c_int(precision), C.c_int(lanes_per_filter)))


class BatchUKF:
    """A batch of independent UKFs resident on one MI355X (opaque ukfb_engine handle)."""

    def __init__(self, model: int, precision: int, capacity: int, device: int = 0, stream=None,
                 lanes_per_filter: int = 0, **cfg):
        """stream:
          "private"  an engine-owned non-blocking stream, what ukfb_create gives a C caller (bench.py: nothing else shares
                     the timed stream; small batches run as split launches on two internal streams, ukfb_config.split_streams);
          "torch"    torch's CURRENT stream on `device` (tensors the caller produces with torch and hands to the "_dev" entry
                     points are then ordered with the engine's launches without any synchronise); raises without torch / a GPU;
          an int     that hipStream_t;
          None       "torch" when torch is imported and sees a GPU, else "private" -- the convenient default of this binding;
                     `stream_kind` ("private" / "torch" / "given") tells which one an engine got."""
        self._lib = load_library()
        self._h = C.c_void_p()
        args = (C.byref(self._h), C.c_int(model), C.c_int(precision), C.c_int64(capacity), C.c_int(device))
        kind = "given"
        if stream is None or stream == "torch":
            ts = _torch_current_stream(device)
            if ts is None and stream == "torch":
                raise UkfbError('stream="torch" needs torch with a visible GPU')
            kind = "torch" if ts is not None else "private"
            stream = ts if ts is not None else "private"
        if stream == "private":
            self.stream_kind = "private"
            _chk(self._lib.ukfb_create(*args, None), "ukfb_create")
        else:
            self.stream_kind = kind
            _chk(self._lib.ukfb_create_on_stream(*args, C.c_void_p(int(stream)) if int(stream) else None), "ukfb_create_on_stream")
        self.model, self.precision, self.capacity, self.device = model, precision, int(capacity), device
        self.S = 13 if model == MODEL_POSE else 14
        self.D = 12 if model == MODEL_POSE else 13
        self.PK = self.D * (self.D + 1) // 2
        self.dtype = np.float64 if precision == F64 else np.float32
        if lanes_per_filter or cfg:
            self.configure(lanes_per_filter=lanes_per_filter, **cfg)

    # ---- lifetime / config
    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._lib.ukfb_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def config(self) -> Config:
        c = Config()
        _chk(self._lib.ukfb_get_config(self._h, C.byref(c)), "ukfb_get_config")
        return c

    def configure(self, **kw):
        c = self.config()
        for k, v in kw.items():
            if k == "lanes_per_filter" and not v:
                continue
            setattr(c, k, v)
        _chk(self._lib.ukfb_set_config(self._h, C.byref(c)), "ukfb_set_config")

    def sync(self):
        _chk(self._lib.ukfb_sync(self._h), "ukfb_sync")

    # ---- state
    def initialize(self, mu, cov, first: int = 0):
        mu = _f64(mu, (-1, self.S)); cov = _f64(cov, (-1, self.D, self.D))
        _chk(self._lib.ukfb_initialize(self._h, C.c_int64(first), C.c_int64(mu.shape[0]), _pd(mu), _pd(cov)),
             "ukfb_initialize")

    def state(self, first: int = 0, count: Optional[int] = None, with_cov: bool = True):
        count = self.capacity - first if count is None else count
        mu = np.empty((count, self.S))
        cov = np.empty((count, self.D, self.D)) if with_cov else None
        init = np.empty(count, dtype=np.uint8)
        _chk(self._lib.ukfb_get_state(self._h, C.c_int64(first), C.c_int64(count), _pd(mu), _pd(cov),
                                      init.ctypes.data_as(C.POINTER(C.c_uint8))), "ukfb_get_state")
        return (mu, cov, init.astype(bool)) if with_cov else (mu, init.astype(bool))

    def status(self, first: int = 0, count: Optional[int] = None):
        count = self.capacity - first if count is None else count
        st = np.empty(count, dtype=np.uint32)
        _chk(self._lib.ukfb_get_status(self._h, C.c_int64(first), C.c_int64(count),
                                       st.ctypes.data_as(C.POINTER(C.c_uint32))), "ukfb_get_status")
        return st

    def status_summary(self) -> int:
        v = C.c_uint32(0)
        _chk(self._lib.ukfb_get_status_summary(self._h, C.byref(v)), "ukfb_get_status_summary")
        return int(v.value)

    def set_last_measurement_time(self, t_us, first: int = 0):
        t = np.ascontiguousarray(t_us, dtype=np.int64)
        _chk(self._lib.ukfb_set_last_measurement_time(self._h, C.c_int64(first), C.c_int64(t.size),
                                                      t.ctypes.data_as(C.POINTER(C.c_int64))), "set_last_time")

    def last_measurement_time(self, first: int = 0, count: Optional[int] = None):
        count = self.capacity - first if count is None else count
        t = np.empty(count, dtype=np.int64)
        _chk(self._lib.ukfb_get_last_measurement_time(self._h, C.c_int64(first), C.c_int64(count),
                                                      t.ctypes.data_as(C.POINTER(C.c_int64))), "get_last_time")
        return t

    def device_views(self):
        mu, cov, st = C.c_void_p(), C.c_void_p(), C.c_void_p()
        _chk(self._lib.ukfb_device_views(self._h, C.byref(mu), C.byref(cov), C.byref(st)), "ukfb_device_views")
        return mu.value, cov.value, st.value

    # ---- noise / inputs
    def set_process_noise(self, R, first: Optional[int] = None):
        R = _f64(R)
        if R.ndim == 2:
            _chk(self._lib.ukfb_set_process_noise(self._h, _pd(R)), "ukfb_set_process_noise")
        else:
            _chk(self._lib.ukfb_set_process_noise_per_filter(self._h, C.c_int64(first or 0), C.c_int64(R.shape[0]),
                                                             _pd(R)), "ukfb_set_process_noise_per_filter")

    def process_noise(self, filter_index: int = 0):
        R = np.empty((self.D, self.D))
        _chk(self._lib.ukfb_get_process_noise(self._h, C.c_int64(filter_index), _pd(R)), "ukfb_get_process_noise")
        return R

    def set_acceleration(self, acc_mu=None, acc_cov=None, first: int = 0):
        am = _f64(acc_mu, (-1, 3)) if acc_mu is not None else None
        ac = _f64(acc_cov, (3, 3)) if acc_cov is not None else None
        n = am.shape[0] if am is not None else 0
        _chk(self._lib.ukfb_pose_set_acceleration(self._h, C.c_int64(first), C.c_int64(n), _pd(am), _pd(ac)),
             "ukfb_pose_set_acceleration")

    def export_body_states(self, first: int = 0, count: Optional[int] = None):
        """BodyStateMeasurement::toRigidBodyState for a range of filters -> [count, 49] records."""
        count = self.capacity - first if count is None else count
        out = np.empty((count, BODY_STATE_SCALARS))
        _chk(self._lib.ukfb_pose_export_body_states(self._h, C.c_int64(first), C.c_int64(count), _pd(out)),
             "ukfb_pose_export_body_states")
        return out

    def import_body_states(self, records, first: int = 0):
        """BodyStateMeasurement::fromRigidBodyState + initializeFilter from [count, 49] records."""
        rec = _f64(records, (-1, BODY_STATE_SCALARS))
        _chk(self._lib.ukfb_pose_import_body_states(self._h, C.c_int64(first), C.c_int64(rec.shape[0]), _pd(rec)),
             "ukfb_pose_import_body_states")

    def bind_acceleration_dev(self, acc_dev):
        _chk(self._lib.ukfb_pose_bind_acceleration_dev(self._h, _devptr(acc_dev)), "ukfb_pose_bind_acceleration_dev")

    def set_orient_params(self, gyro_bias_tau: float, acc_bias_tau: float, earth_rotation):
        er = _f64(earth_rotation, (3,))
        _chk(self._lib.ukfb_orient_set_params(self._h, C.c_double(gyro_bias_tau), C.c_double(acc_bias_tau), _pd(er)),
             "ukfb_orient_set_params")

    def set_orient_inputs(self, gyro=None, acc=None, first: int = 0):
        g = _f64(gyro, (-1, 3)) if gyro is not None else None
        a = _f64(acc, (-1, 3)) if acc is not None else None
        n = g.shape[0] if g is not None else (a.shape[0] if a is not None else 0)
        _chk(self._lib.ukfb_orient_set_inputs(self._h, C.c_int64(first), C.c_int64(n), _pd(g), _pd(a)),
             "ukfb_orient_set_inputs")

    def bind_orient_inputs_dev(self, gyro_dev, acc_dev):
        _chk(self._lib.ukfb_orient_bind_inputs_dev(self._h, _devptr(gyro_dev), _devptr(acc_dev)),
             "ukfb_orient_bind_inputs_dev")

    def rotation_rate(self, first: int = 0, count: Optional[int] = None):
        count = self.capacity - first if count is None else count
        out = np.empty((count, 3))
        _chk(self._lib.ukfb_orient_get_rotation_rate(self._h, C.c_int64(first), C.c_int64(count), _pd(out)),
             "ukfb_orient_get_rotation_rate")
        return out

    # ---- predict
    def predict(self, dt):
        if np.isscalar(dt):
            _chk(self._lib.ukfb_predict(self._h, C.c_double(float(dt))), "ukfb_predict")
        else:
            d = _f64(dt, (self.capacity,))
            _chk(self._lib.ukfb_predict_dt(self._h, _pd(d)), "ukfb_predict_dt")

    def predict_timestamps(self, ts_us):
        t = np.ascontiguousarray(ts_us, dtype=np.int64).reshape(self.capacity)
        _chk(self._lib.ukfb_predict_timestamps(self._h, t.ctypes.data_as(C.POINTER(C.c_int64))),
             "ukfb_predict_timestamps")

    def predict_dt_dev(self, dt_dev):
        _chk(self._lib.ukfb_predict_dt_dev(self._h, _devptr(dt_dev)), "ukfb_predict_dt_dev")

    def predict_timestamps_dev(self, ts_dev):
        _chk(self._lib.ukfb_predict_timestamps_dev(self._h, _devptr(ts_dev)), "ukfb_predict_timestamps_dev")

    # ---- update
    def update(self, meas_model, z, Q, active=None):
        z = _f64(z, (self.capacity, 3)); Q = _f64(Q, (self.capacity, 3, 3))
        if np.isscalar(meas_model):
            act = np.ascontiguousarray(active, dtype=np.uint8) if active is not None else None
            _chk(self._lib.ukfb_update(self._h, C.c_int(int(meas_model)), _pd(z), _pd(Q),
                                       act.ctypes.data_as(C.POINTER(C.c_uint8)) if act is not None else None),
                 "ukfb_update")
        else:
            m = np.ascontiguousarray(meas_model, dtype=np.int32).reshape(self.capacity)
            _chk(self._lib.ukfb_update_mixed(self._h, m.ctypes.data_as(C.POINTER(C.c_int32)), _pd(z), _pd(Q)),
                 "ukfb_update_mixed")

    def update_dev(self, meas_model_uniform: int, z_dev, Q_dev, meas_model_dev=None):
        _chk(self._lib.ukfb_update_dev(self._h, C.c_int(meas_model_uniform), _devptr(meas_model_dev), _devptr(z_dev),
                                       _devptr(Q_dev)), "ukfb_update_dev")

    # ---- fused cycle
    def cycle(self, dt: float, meas_model: int, z, Q):
        z = _f64(z, (self.capacity, 3)); Q = _f64(Q, (self.capacity, 3, 3))
        _chk(self._lib.ukfb_cycle(self._h, C.c_double(dt), C.c_int(meas_model), _pd(z), _pd(Q)), "ukfb_cycle")

    def cycle_uniform_q(self, dt: float, meas_model: int, z, Q9):
        """fused cycle with ONE 3x3 measurement covariance for the whole batch (host arrays)"""
        z = _f64(z, (self.capacity, 3)); Q9 = _f64(Q9, (9,))
        _chk(self._lib.ukfb_cycle_uniform_q(self._h, C.c_double(dt), C.c_int(meas_model), _pd(z), _pd(Q9)), "ukfb_cycle_uniform_q")

    def cycle_uniform_q_dev(self, dt: float, meas_model: int, z_dev, Q9_dev):
        _chk(self._lib.ukfb_cycle_uniform_q_dev(self._h, C.c_double(dt), C.c_int(meas_model), _devptr(z_dev), _devptr(Q9_dev)),
             "ukfb_cycle_uniform_q_dev")

    def update_uniform_q(self, meas_model: int, z, Q9, active=None):
        z = _f64(z, (self.capacity, 3)); Q9 = _f64(Q9, (9,))
        a = None if active is None else np.ascontiguousarray(active, dtype=np.uint8).reshape(self.capacity)
        _chk(self._lib.ukfb_update_uniform_q(self._h, C.c_int(meas_model), _pd(z), _pd(Q9),
                                             a.ctypes.data_as(C.POINTER(C.c_uint8)) if a is not None else None), "ukfb_update_uniform_q")

    def cycle_dev(self, dt: float, meas_model_uniform: int, z_dev, Q_dev, meas_model_dev=None):
        _chk(self._lib.ukfb_cycle_dev(self._h, C.c_double(dt), C.c_int(meas_model_uniform), _devptr(meas_model_dev),
                                      _devptr(z_dev), _devptr(Q_dev)), "ukfb_cycle_dev")

    def cycle_multi_dev(self, cycles: int, dt: float, meas_model: int, z_dev, Q_dev, slots: int, first_slot: int = 0,
                        in_a_dev=None, in_b_dev=None):
        """`cycles` fused cycles in one launch, the filters stay in LDS in between; cycle c reads slot (first_slot + c) % slots
        of the device rings z_dev [slots][capacity][3], Q_dev [slots][capacity][9] and, if given, in_a_dev / in_b_dev
        [slots][capacity][3] (Pose: acceleration; Orient: acceleration, rotation rate) in place of the latched inputs."""
        _chk(self._lib.ukfb_cycle_multi_dev(self._h, C.c_int(cycles), C.c_double(dt), C.c_int(meas_model), C.c_int(slots),
                                            C.c_int(first_slot), _devptr(in_a_dev), _devptr(in_b_dev), _devptr(z_dev),
                                            _devptr(Q_dev)), "ukfb_cycle_multi_dev")

    def cycle_multi_mixed_dev(self, cycles: int, dt: float, meas_model_dev, z_dev, Q_dev, slots: int, first_slot: int = 0,
                              in_a_dev=None, in_b_dev=None):
        """cycle_multi_dev with per-filter model ids per cycle: meas_model_dev int32 [slots][capacity], negative = none."""
        _chk(self._lib.ukfb_cycle_multi_mixed_dev(self._h, C.c_int(cycles), C.c_double(dt), C.c_int(slots), C.c_int(first_slot),
                                                  _devptr(in_a_dev), _devptr(in_b_dev), _devptr(meas_model_dev),
                                                  _devptr(z_dev), _devptr(Q_dev)), "ukfb_cycle_multi_mixed_dev")

    def cycle_schedule_dev(self, dt, meas_model, z_dev, Q_dev, slots: int, first_slot: int = 0, in_a_dev=None, in_b_dev=None):
        """Scheduled multi-cycle launch: cycle c predicts by dt[c] and updates with model meas_model[c] (negative: prediction
        only); inputs from the device rings as in cycle_multi_dev."""
        d = np.ascontiguousarray(dt, dtype=np.float64).reshape(-1)
        m = np.ascontiguousarray(meas_model, dtype=np.int32).reshape(-1)
        if d.size != m.size:
            raise ValueError("dt and meas_model need one entry per cycle")
        _chk(self._lib.ukfb_cycle_schedule_dev(self._h, C.c_int(d.size), _pd(d), m.ctypes.data_as(C.POINTER(C.c_int32)),
                                               C.c_int(slots), C.c_int(first_slot), _devptr(in_a_dev), _devptr(in_b_dev),
                                               _devptr(z_dev), _devptr(Q_dev)), "ukfb_cycle_schedule_dev")

    def cycle_multi(self, dt: float, meas_model: int, z, Q, in_a=None, in_b=None):
        """Host arrays, one input set per cycle: z [cycles, capacity, 3], Q [cycles, capacity, 3, 3], in_a / in_b
        [cycles, capacity, 3] or None (the latched inputs); all cycles in one launch."""
        z = np.ascontiguousarray(z, dtype=np.float64)
        cycles = z.shape[0]
        z = _f64(z, (cycles, self.capacity, 3)); Q = _f64(Q, (cycles, self.capacity, 3, 3))
        a = None if in_a is None else _f64(in_a, (cycles, self.capacity, 3))
        b = None if in_b is None else _f64(in_b, (cycles, self.capacity, 3))
        _chk(self._lib.ukfb_cycle_multi(self._h, C.c_int(cycles), C.c_double(dt), C.c_int(meas_model),
                                        _pd(a) if a is not None else None, _pd(b) if b is not None else None, _pd(z), _pd(Q)),
             "ukfb_cycle_multi")

    def cycle_timestamps(self, ts_us, meas_model, z, Q):
        """Fused predictionStepFromSampleTime(ts[i]) + integrateMeasurement(model[i]); ts < 0: no sample."""
        t = np.ascontiguousarray(ts_us, dtype=np.int64).reshape(self.capacity)
        m = np.ascontiguousarray(meas_model, dtype=np.int32).reshape(self.capacity)
        z = _f64(z, (self.capacity, 3)); Q = _f64(Q, (self.capacity, 3, 3))
        _chk(self._lib.ukfb_cycle_timestamps(self._h, t.ctypes.data_as(C.POINTER(C.c_int64)),
                                             m.ctypes.data_as(C.POINTER(C.c_int32)), _pd(z), _pd(Q)),
             "ukfb_cycle_timestamps")

    def process_events(self, filter_index, ts_us, meas_model, z, Q):
        """Time-ordered asynchronous measurement stream (any arrival order).  Returns (status_or, rounds)."""
        f = np.ascontiguousarray(filter_index, dtype=np.int64).reshape(-1)
        n = f.size
        t = np.ascontiguousarray(ts_us, dtype=np.int64).reshape(n)
        m = np.ascontiguousarray(meas_model, dtype=np.int32).reshape(n)
        z = _f64(z, (n, 3)); Q = _f64(Q, (n, 3, 3))
        st, rounds = C.c_uint32(0), C.c_int64(0)
        _chk(self._lib.ukfb_process_events(self._h, C.c_int64(n), f.ctypes.data_as(C.POINTER(C.c_int64)),
                                           t.ctypes.data_as(C.POINTER(C.c_int64)),
                                           m.ctypes.data_as(C.POINTER(C.c_int32)), _pd(z), _pd(Q), C.byref(st),
                                           C.byref(rounds)), "ukfb_process_events")
        return int(st.value), int(rounds.value)

    def process_events_dev(self, n_events: int, filter_dev, ts_us_dev, meas_model_dev, z_dev, Q_dev):
        """The same stream already resident in HBM (int64, int64, int32, z / Q in the engine's precision)."""
        st, rounds = C.c_uint32(0), C.c_int64(0)
        _chk(self._lib.ukfb_process_events_dev(self._h, C.c_int64(int(n_events)), _devptr(filter_dev), _devptr(ts_us_dev),
                                               _devptr(meas_model_dev), _devptr(z_dev), _devptr(Q_dev), C.byref(st),
                                               C.byref(rounds)), "ukfb_process_events_dev")
        return int(st.value), int(rounds.value)

    # ---- measurement of the engine
    def last_launch_info(self):
        name = C.create_string_buffer(256)
        lds, fpw, grid = C.c_int(0), C.c_int(0), C.c_int64(0)
        _chk(self._lib.ukfb_last_launch_info(self._h, name, C.c_int(256), C.byref(lds), C.byref(fpw), C.byref(grid)),
             "ukfb_last_launch_info")
        return {"kernel": name.value.decode(), "lds_bytes": lds.value, "filters_per_workgroup": fpw.value,
                "grid": grid.value}

    def last_model_groups(self):
        """int32 list of the most recent launch that grouped its filters by update class (-1 = padding); see ukf_batch.h"""
        items = C.c_int64(0)
        cap = int(self.capacity) + 16
        out = np.empty(cap, dtype=np.int32)
        _chk(self._lib.ukfb_last_model_groups(self._h, out.ctypes.data_as(C.POINTER(C.c_int32)), C.c_int64(cap), C.byref(items)),
             "ukfb_last_model_groups")
        return out[:int(items.value)].copy()

    def timer_begin(self):
        _chk(self._lib.ukfb_timer_begin(self._h), "ukfb_timer_begin")

    def timer_end(self) -> float:
        ms = C.c_float(0)
        _chk(self._lib.ukfb_timer_end(self._h, C.byref(ms)), "ukfb_timer_end")
        return float(ms.value)


class BatchPoseUKF(BatchUKF):
    """Batched sibling of pose_estimation::PoseUKF (pose_with_velocity/PoseUKF.hpp:20-96)."""

    def __init__(self, capacity: int, precision: int = F64, device: int = 0, **kw):
        super().__init__(MODEL_POSE, precision, capacity, device, **kw)
        # PoseUKF ctor defaults (PoseUKF.cpp:103-107)
        self.set_process_noise(np.diag([0.01] * 3 + [0.001] * 3 + [0.00001] * 3 + [0.00001] * 3))


class BatchOrientationUKF(BatchUKF):
    """Batched sibling of pose_estimation::OrientationUKF (orientation_estimator/OrientationUKF.hpp:20-62)."""

    EARTHW = (2.0 * np.pi) / 86164.0  # GravitationalModel.hpp:16

    def __init__(self, capacity: int, gyro_bias_tau: float, acc_bias_tau: float, latitude: float,
                 precision: int = F64, device: int = 0, **kw):
        super().__init__(MODEL_ORIENT, precision, capacity, device, **kw)
        # OrientationUKF.cpp:47
        self.earth_rotation = np.array([self.EARTHW * np.cos(latitude), 0.0, self.EARTHW * np.sin(latitude)])
        self.set_orient_params(gyro_bias_tau, acc_bias_tau, self.earth_rotation)

    def initialize(self, mu, cov, first: int = 0):
        """initializeFilter plus the ctor's input latches (OrientationUKF.cpp:49-50)."""
        mu = _f64(mu, (-1, self.S))
        super().initialize(mu, cov, first)
        n = mu.shape[0]
        acc = np.zeros((n, 3)); acc[:, 2] = mu[:, 13]
        self.set_orient_inputs(gyro=np.zeros((n, 3)), acc=acc, first=first)


class _ShardView(BatchUKF):
    """A shard's engine seen through the per-engine binding (the group owns the handle: close() is a no-op)."""

    def __init__(self, lib, handle, model, precision, capacity, device):   # noqa: D401 (no ukfb_create here)
        self._lib, self._h = lib, handle
        self.model, self.precision, self.capacity, self.device = model, precision, int(capacity), device
        self.S = 13 if model == MODEL_POSE else 14
        self.D = 12 if model == MODEL_POSE else 13
        self.PK = self.D * (self.D + 1) // 2
        self.dtype = np.float64 if precision == F64 else np.float32
        self.stream_kind = "private"

    def close(self):
        self._h = C.c_void_p()


class UKFGroup:
    """One process, several GPUs: `total` independent filters in contiguous shards, one engine per device
    (ukfb_group_* of include/ukf_batch.h).  Mirrors pose_estimation::ShardedBatchPoseUKF (include/pose_estimation/Batch.hpp)."""

    def __init__(self, model: int, precision: int, total: int, devices):
        self._lib = load_library()
        self._g = C.c_void_p()
        devs = (C.c_int * len(devices))(*[int(d) for d in devices])
        _chk(self._lib.ukfb_group_create(C.byref(self._g), C.c_int(model), C.c_int(precision), C.c_int64(total), devs,
                                         C.c_int(len(devices))), "ukfb_group_create")
        self.model, self.precision, self.total = model, precision, int(total)
        self.S = 13 if model == MODEL_POSE else 14
        self.D = 12 if model == MODEL_POSE else 13
        self.n = int(self._lib.ukfb_group_size(self._g))
        self.shards = []
        for r in range(self.n):
            h, dev, first, count = C.c_void_p(), C.c_int(0), C.c_int64(0), C.c_int64(0)
            _chk(self._lib.ukfb_group_shard(self._g, C.c_int(r), C.byref(h), C.byref(dev), C.byref(first), C.byref(count)),
                 "ukfb_group_shard")
            self.shards.append({"engine": _ShardView(self._lib, h, model, precision, count.value, dev.value),
                                "device": dev.value, "first": first.value, "count": count.value})
        if model == MODEL_POSE:   # PoseUKF ctor defaults (PoseUKF.cpp:103-107), as BatchPoseUKF
            self.set_process_noise(np.diag([0.01] * 3 + [0.001] * 3 + [0.00001] * 3 + [0.00001] * 3))

    def close(self):
        if getattr(self, "_g", None) and self._g.value:
            self._lib.ukfb_group_destroy(self._g)
            self._g = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ptrs(self, xs):
        if xs is None:
            return None
        assert len(xs) == self.n, "one device pointer per shard"
        return (C.c_void_p * self.n)(*[(_devptr(x).value if x is not None else None) for x in xs])

    def configure(self, **kw):
        c = self.shards[0]["engine"].config()
        for k, v in kw.items():
            setattr(c, k, v)
        _chk(self._lib.ukfb_group_set_config(self._g, C.byref(c)), "ukfb_group_set_config")

    def initialize(self, mu, cov, first: int = 0):
        mu = _f64(mu, (-1, self.S)); cov = _f64(cov, (-1, self.D, self.D))
        _chk(self._lib.ukfb_group_initialize(self._g, C.c_int64(first), C.c_int64(mu.shape[0]), _pd(mu), _pd(cov)),
             "ukfb_group_initialize")
        if self.model == MODEL_ORIENT:   # the reference constructor's input latches (OrientationUKF.cpp:49-50), as BatchOrientationUKF
            n = mu.shape[0]
            acc = np.zeros((n, 3)); acc[:, 2] = mu[:, 13]
            self.set_orient_inputs(gyro=np.zeros((n, 3)), acc=acc, first=first)

    def state(self, first: int = 0, count: Optional[int] = None):
        count = self.total - first if count is None else count
        mu = np.empty((count, self.S)); cov = np.empty((count, self.D, self.D)); init = np.empty(count, dtype=np.uint8)
        _chk(self._lib.ukfb_group_get_state(self._g, C.c_int64(first), C.c_int64(count), _pd(mu), _pd(cov),
                                            init.ctypes.data_as(C.POINTER(C.c_uint8))), "ukfb_group_get_state")
        return mu, cov, init.astype(bool)

    def status(self, first: int = 0, count: Optional[int] = None):
        count = self.total - first if count is None else count
        st = np.empty(count, dtype=np.uint32)
        _chk(self._lib.ukfb_group_get_status(self._g, C.c_int64(first), C.c_int64(count), st.ctypes.data_as(C.POINTER(C.c_uint32))),
             "ukfb_group_get_status")
        return st

    def status_summary(self) -> int:
        v = C.c_uint32(0)
        _chk(self._lib.ukfb_group_get_status_summary(self._g, C.byref(v)), "ukfb_group_get_status_summary")
        return int(v.value)

    def set_process_noise(self, R):
        _chk(self._lib.ukfb_group_set_process_noise(self._g, _pd(_f64(R, (self.D, self.D)))), "ukfb_group_set_process_noise")

    def set_acceleration(self, acc_mu=None, acc_cov=None, first: int = 0):
        am = _f64(acc_mu, (-1, 3)) if acc_mu is not None else None
        ac = _f64(acc_cov, (3, 3)) if acc_cov is not None else None
        _chk(self._lib.ukfb_group_pose_set_acceleration(self._g, C.c_int64(first), C.c_int64(am.shape[0] if am is not None else 0),
                                                        _pd(am), _pd(ac)), "ukfb_group_pose_set_acceleration")

    def set_orient_params(self, gyro_bias_tau: float, acc_bias_tau: float, earth_rotation):
        _chk(self._lib.ukfb_group_orient_set_params(self._g, C.c_double(gyro_bias_tau), C.c_double(acc_bias_tau),
                                                    _pd(_f64(earth_rotation, (3,)))), "ukfb_group_orient_set_params")

    def set_orient_inputs(self, gyro=None, acc=None, first: int = 0):
        g = _f64(gyro, (-1, 3)) if gyro is not None else None
        a = _f64(acc, (-1, 3)) if acc is not None else None
        n = g.shape[0] if g is not None else (a.shape[0] if a is not None else 0)
        _chk(self._lib.ukfb_group_orient_set_inputs(self._g, C.c_int64(first), C.c_int64(n), _pd(g), _pd(a)),
             "ukfb_group_orient_set_inputs")

    def predict(self, dt: float):
        _chk(self._lib.ukfb_group_predict(self._g, C.c_double(dt)), "ukfb_group_predict")

    def update(self, meas_model: int, z, Q):
        z = _f64(z, (self.total, 3)); Q = _f64(Q, (self.total, 3, 3))
        _chk(self._lib.ukfb_group_update(self._g, C.c_int(meas_model), _pd(z), _pd(Q)), "ukfb_group_update")

    def cycle(self, dt: float, meas_model: int, z, Q):
        z = _f64(z, (self.total, 3)); Q = _f64(Q, (self.total, 3, 3))
        _chk(self._lib.ukfb_group_cycle(self._g, C.c_double(dt), C.c_int(meas_model), _pd(z), _pd(Q)), "ukfb_group_cycle")

    def bind_acceleration_dev(self, acc_devs):
        _chk(self._lib.ukfb_group_pose_bind_acceleration_dev(self._g, self._ptrs(acc_devs)), "ukfb_group_pose_bind_acceleration_dev")

    def bind_orient_inputs_dev(self, gyro_devs, acc_devs):
        _chk(self._lib.ukfb_group_orient_bind_inputs_dev(self._g, self._ptrs(gyro_devs), self._ptrs(acc_devs)),
             "ukfb_group_orient_bind_inputs_dev")

    def cycle_dev(self, dt: float, meas_model: int, z_devs, Q_devs):
        _chk(self._lib.ukfb_group_cycle_dev(self._g, C.c_double(dt), C.c_int(meas_model), self._ptrs(z_devs), self._ptrs(Q_devs)),
             "ukfb_group_cycle_dev")

    def cycle_multi_dev(self, cycles: int, dt: float, meas_model: int, z_devs, Q_devs, slots: int, first_slot: int = 0,
                        in_a_devs=None, in_b_devs=None):
        _chk(self._lib.ukfb_group_cycle_multi_dev(self._g, C.c_int(cycles), C.c_double(dt), C.c_int(meas_model), C.c_int(slots),
                                                  C.c_int(first_slot), self._ptrs(in_a_devs), self._ptrs(in_b_devs),
                                                  self._ptrs(z_devs), self._ptrs(Q_devs)), "ukfb_group_cycle_multi_dev")

    def cycle_mixed_dev(self, dt: float, meas_model_devs, z_devs, Q_devs):
        """per-filter model ids resident on the devices (int32, one array per shard)"""
        _chk(self._lib.ukfb_group_cycle_mixed_dev(self._g, C.c_double(dt), self._ptrs(meas_model_devs), self._ptrs(z_devs),
                                                  self._ptrs(Q_devs)), "ukfb_group_cycle_mixed_dev")

    def cycle_timestamps(self, ts_us, meas_model, z, Q):
        t = np.ascontiguousarray(ts_us, dtype=np.int64).reshape(self.total)
        m = np.ascontiguousarray(meas_model, dtype=np.int32).reshape(self.total)
        z = _f64(z, (self.total, 3)); Q = _f64(Q, (self.total, 3, 3))
        _chk(self._lib.ukfb_group_cycle_timestamps(self._g, t.ctypes.data_as(C.POINTER(C.c_int64)),
                                                   m.ctypes.data_as(C.POINTER(C.c_int32)), _pd(z), _pd(Q)), "ukfb_group_cycle_timestamps")

    def process_events(self, filter_index, ts_us, meas_model, z, Q):
        """Time-ordered asynchronous stream over the sharded batch (filter indices in batch numbering, any arrival order).
        Returns (status_or, rounds of the shard that needed most)."""
        f = np.ascontiguousarray(filter_index, dtype=np.int64).reshape(-1)
        n = f.size
        t = np.ascontiguousarray(ts_us, dtype=np.int64).reshape(n)
        m = np.ascontiguousarray(meas_model, dtype=np.int32).reshape(n)
        z = _f64(z, (n, 3)); Q = _f64(Q, (n, 3, 3))
        st, rounds = C.c_uint32(0), C.c_int64(0)
        _chk(self._lib.ukfb_group_process_events(self._g, C.c_int64(n), f.ctypes.data_as(C.POINTER(C.c_int64)),
                                                 t.ctypes.data_as(C.POINTER(C.c_int64)), m.ctypes.data_as(C.POINTER(C.c_int32)),
                                                 _pd(z), _pd(Q), C.byref(st), C.byref(rounds)), "ukfb_group_process_events")
        return int(st.value), int(rounds.value)

    def sync(self):
        _chk(self._lib.ukfb_group_sync(self._g), "ukfb_group_sync")

    def timer_begin(self):
        _chk(self._lib.ukfb_group_timer_begin(self._g), "ukfb_group_timer_begin")

    def timer_end(self):
        """(slowest shard's ms, [ms per shard])"""
        mx = C.c_float(0)
        per = (C.c_float * self.n)()
        _chk(self._lib.ukfb_group_timer_end(self._g, C.byref(mx), per), "ukfb_group_timer_end")
        return float(mx.value), [float(x) for x in per]

    def gather_means(self, out_devs):
        """RCCL all-gather of the means: out_devs[r] = device buffer [total][S] (engine precision) on shard r's device."""
        _chk(self._lib.ukfb_group_gather_means(self._g, self._ptrs(out_devs)), "ukfb_group_gather_means")

    def last_gather_exchange(self) -> str:
        """which exchange the last gather_means used: "rccl" (one shard per device) or "copies" (shards sharing a device)"""
        return {0: "none", 1: "rccl", 2: "copies"}.get(int(self._lib.ukfb_group_last_gather_exchange(self._g)), "?")
