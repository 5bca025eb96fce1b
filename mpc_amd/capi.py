"""ctypes binding of libmpc_amd.so (the HIP product library) — the Python caller's view of
include/mpc_batch.h.  There is NO CPU fallback: if the library is missing or a call fails,
this module raises.

`BatchSolver` is to the C-ABI what a list of acados capsules plus
`mobile_robot_acados_batch_solve` (src/c_generated_code/acados_solver_mobile_robot.c:951-961)
is to the reference.
"""
import ctypes
import os

import numpy as np

from .variants import OcpSpec

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MPC_AMD_LIB") or os.path.join(_HERE, "libmpc_amd.so")  # override: A/B builds in development

_dp = ctypes.POINTER(ctypes.c_double)
_ip = ctypes.POINTER(ctypes.c_int)
_vp = ctypes.c_void_p

EXPORTS = (
    "mpc_spec_variant", "mpc_batch_create", "mpc_batch_solve", "mpc_batch_solve_device",
    "mpc_batch_reset", "mpc_batch_get", "mpc_batch_sim_step", "mpc_batch_free",
    "mpc_batch_last_kernel_ms", "mpc_batch_horizon", "mpc_batch_capacity",
    "mpc_batch_last_error", "mpc_batch_version", "mpc_batch_debug_linearize", "mpc_batch_device_bytes", "mpc_batch_last_phase0_ms",
    "mpc_batch_get_sqp", "mpc_batch_track_set_path", "mpc_batch_track_init", "mpc_batch_track_run", "mpc_batch_set_model", "mpc_batch_get_diag",
    "mpc_batch_shift", "mpc_batch_track_set_shift", "mpc_batch_solve_static",
)


class MpcError(RuntimeError):
    pass


_lib = None


def _preload_torch_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64.so
    (SONAME libamdhip64.so.7, the same as /opt/rocm's).  If libmpc_amd.so pulled in
    /opt/rocm's copy first, a later `import torch` would load a second runtime and find no
    GPU.  Loading torch's copy first (by path, without importing torch) makes our DT_NEEDED
    resolve to it by SONAME; without torch installed the system runtime is used."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    path = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(path):
        ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)


def lib():
    """Load libmpc_amd.so (built by `make -C mpc_amd/csrc` / __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MpcError(f"{LIB_PATH} is missing: build it with `make -C mpc_amd/csrc` "
                       "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    _preload_torch_hip_runtime()
    L = ctypes.CDLL(LIB_PATH)
    L.mpc_spec_variant.argtypes = [ctypes.c_char_p, ctypes.POINTER(OcpSpec)]
    L.mpc_batch_create.argtypes = [ctypes.POINTER(_vp), ctypes.POINTER(OcpSpec), ctypes.c_int, _ip, ctypes.c_int]
    L.mpc_batch_solve.argtypes = [_vp, ctypes.c_int] + [_vp] * 8     # (pointers through _fp: the batch = 1 loop)
    L.mpc_batch_solve_device.argtypes = [_vp, ctypes.c_int] + [_vp] * 8 + [_vp]
    L.mpc_batch_reset.argtypes = [_vp, ctypes.POINTER(ctypes.c_ubyte)]
    L.mpc_batch_get.argtypes = [_vp, ctypes.c_int, _dp, _dp, _dp]
    L.mpc_batch_sim_step.argtypes = [_vp, ctypes.c_int, _dp, _dp]
    L.mpc_batch_free.argtypes = [_vp]
    L.mpc_batch_free.restype = None
    L.mpc_batch_last_kernel_ms.argtypes = [_vp]
    L.mpc_batch_last_kernel_ms.restype = ctypes.c_double
    L.mpc_batch_debug_linearize.argtypes = [_vp, ctypes.c_int, _dp, _dp, _dp]
    L.mpc_batch_device_bytes.argtypes = [_vp]
    L.mpc_batch_device_bytes.restype = ctypes.c_size_t
    L.mpc_batch_last_phase0_ms.argtypes = [_vp]
    L.mpc_batch_last_phase0_ms.restype = ctypes.c_double
    L.mpc_batch_get_sqp.argtypes = [_vp, ctypes.c_int, _ip, _dp]
    L.mpc_batch_track_set_path.argtypes = [_vp, ctypes.c_int, _dp, ctypes.c_int, _dp]
    L.mpc_batch_track_init.argtypes = [_vp, ctypes.c_int, _dp, ctypes.c_double, ctypes.c_double]
    L.mpc_batch_track_run.argtypes = [_vp, ctypes.c_int, ctypes.c_int, _dp, _dp, _ip, _ip]
    L.mpc_batch_set_model.argtypes = [_vp, ctypes.c_int]
    L.mpc_batch_shift.argtypes = [_vp, ctypes.POINTER(ctypes.c_ubyte)]
    L.mpc_batch_track_set_shift.argtypes = [_vp, ctypes.c_int]
    L.mpc_batch_get_diag.argtypes = [_vp, ctypes.c_int, _ip]
    L.mpc_batch_solve_static.argtypes = [ctypes.POINTER(OcpSpec), ctypes.c_int, ctypes.c_int] + [_vp] * 6 + [_vp]
    L.mpc_batch_horizon.argtypes = [_vp]
    L.mpc_batch_capacity.argtypes = [_vp]
    L.mpc_batch_last_error.argtypes = [_vp]
    L.mpc_batch_last_error.restype = ctypes.c_char_p
    L.mpc_batch_version.restype = ctypes.c_char_p
    _lib = L
    return L


def _p(a):
    return None if a is None else a.ctypes.data_as(_dp)


def _pi(a):
    return None if a is None else a.ctypes.data_as(_ip)


def _fp(a):
    """Pointer to a numpy array's data for a c_void_p parameter: a.ctypes.data_as() costs 2-4 us per
    array, seven of them were a tenth of a batch = 1 tick."""
    if a is None:
        return None
    try:
        return ctypes.byref(ctypes.c_char.from_buffer(a))
    except (TypeError, BufferError, ValueError):      # read-only or empty arrays
        return a.ctypes.data


def spec_variant(name):
    s = OcpSpec()
    if lib().mpc_spec_variant(name.encode(), ctypes.byref(s)) != 0:
        raise MpcError(f"unknown variant {name!r}")
    return s


STATIC_MAX_LINES, STATIC_MAX_B = 524288, 262144


def solve_static(spec, B, cold, d_x0, d_yref, d_yref_e, d_u0, d_status=0, d_qp_iter=0, stream=0):
    """mpc_batch_solve_static: the SQP-RTI step on the library's static capsules (no handle, no
    allocation); raw device pointers (ints); asynchronous.  Returns the C return code."""
    return lib().mpc_batch_solve_static(ctypes.byref(spec), B, 1 if cold else 0, d_x0, d_yref, d_yref_e, d_u0,
                                        d_status or None, d_qp_iter or None, stream or None)


class BatchSolver:
    """A handle of `capacity` independent OCP capsules on one or several GPUs."""

    def __init__(self, spec, capacity, devices=None):
        self._h = _vp()
        self.spec = spec
        self.N = int(spec.N)
        self.capacity = int(capacity)
        L = lib()
        if devices:
            arr = (ctypes.c_int * len(devices))(*devices)
            rc = L.mpc_batch_create(ctypes.byref(self._h), ctypes.byref(spec), capacity, arr, len(devices))
        else:
            rc = L.mpc_batch_create(ctypes.byref(self._h), ctypes.byref(spec), capacity, None, 0)
        if rc != 0:
            self._h = _vp()
            raise MpcError(f"mpc_batch_create failed with {rc} "
                           "(-1 bad argument, -2 HIP error, -3 no GPU visible)")

    def _check(self, rc, what):
        if rc != 0:
            msg = lib().mpc_batch_last_error(self._h)
            raise MpcError(f"{what} failed with {rc}: {msg.decode() if msg else ''}")

    def solve(self, x0, yref, yref_e, xtraj=None, utraj=None, u0_out=None):
        """One SQP-RTI step per instance.  Host numpy arrays in the reference's AoS layouts.
        Passing xtraj/utraj loads them as the iterate first (and returns the new iterate);
        otherwise the handle's internal iterate is used, like an acados capsule.  Arrays that sit
        in pinned host memory (e.g. views of torch tensors after .pin_memory()) are read in place
        by the kernels; u0_out ([B,2] float64, C-contiguous) receives u0 instead of a new array."""
        B, N = x0.shape[0], self.N
        x0 = np.ascontiguousarray(x0, np.float64)
        yref = np.ascontiguousarray(yref, np.float64)
        yref_e = np.ascontiguousarray(yref_e, np.float64)
        if x0.shape != (B, 3) or yref.shape != (B, N, 5) or yref_e.shape != (B, 3):
            raise ValueError("expected x0 [B,3], yref [B,N,5], yref_e [B,3]")
        xt = ut = None
        if xtraj is not None or utraj is not None:
            xt = np.ascontiguousarray(xtraj, np.float64).copy()
            ut = np.ascontiguousarray(utraj, np.float64).copy()
            if xt.shape != (B, N + 1, 3) or ut.shape != (B, N, 2):
                raise ValueError("expected xtraj [B,N+1,3], utraj [B,N,2]")
        if u0_out is None:
            u0 = np.zeros((B, 2))
        else:
            u0 = u0_out
            if u0.shape != (B, 2) or u0.dtype != np.float64 or not u0.flags.c_contiguous:
                raise ValueError("expected u0_out [B,2] float64, C-contiguous")
        status = np.zeros(B, np.int32)
        it = np.zeros(B, np.int32)
        rc = lib().mpc_batch_solve(self._h, B, _fp(x0), _fp(yref), _fp(yref_e), _fp(u0), _fp(xt), _fp(ut), _fp(status), _fp(it))
        self._check(rc, "mpc_batch_solve")
        return dict(u0=u0, xtraj=xt, utraj=ut, status=status, qp_iter=it)

    def solve_device(self, B, d_x0, d_yref, d_yref_e, d_u0=0, d_xtraj=0, d_utraj=0, d_status=0, d_qp_iter=0, stream=0):
        """Same call on raw device pointers (ints, e.g. torch.Tensor.data_ptr()); asynchronous."""
        rc = lib().mpc_batch_solve_device(self._h, B, d_x0, d_yref, d_yref_e, d_u0 or None, d_xtraj or None,
                                          d_utraj or None, d_status or None, d_qp_iter or None, stream or None)
        self._check(rc, "mpc_batch_solve_device")

    def get(self, B=None, want_pi=True):
        B = self.capacity if B is None else B
        N = self.N
        xt, ut = np.zeros((B, N + 1, 3)), np.zeros((B, N, 2))
        pi = np.zeros((B, N, 3)) if want_pi else None
        self._check(lib().mpc_batch_get(self._h, B, _p(xt), _p(ut), _p(pi)), "mpc_batch_get")
        return xt, ut, pi

    def get_sqp(self, B=None):
        """Full SQP: (sqp_iter [B], residuals [B,4] = stat, eq, ineq, comp) of the last solve."""
        B = self.capacity if B is None else B
        it, res = np.zeros(B, np.int32), np.zeros((B, 4))
        self._check(lib().mpc_batch_get_sqp(self._h, B, _pi(it), _p(res)), "mpc_batch_get_sqp")
        return it, res

    # ---- closed loop on the device: B `Optimizer`s following one shared path ----
    def track_set_path(self, state_refs, input_refs):
        s = np.ascontiguousarray(state_refs, np.float64)
        u = np.ascontiguousarray(input_refs, np.float64)
        if s.ndim != 2 or s.shape[1] != 3 or u.ndim != 2 or u.shape[1] != 2:
            raise ValueError("expected state_refs [M,3], input_refs [Mi,2]")
        self._check(lib().mpc_batch_track_set_path(self._h, s.shape[0], _p(s), u.shape[0], _p(u)), "mpc_batch_track_set_path")

    def track_init(self, x_init, T, v_ref):
        x = np.ascontiguousarray(x_init, np.float64)
        if x.ndim != 2 or x.shape[1] != 3 or x.shape[0] > self.capacity:
            raise ValueError("expected x_init [B,3], B <= capacity")
        self._track_B = x.shape[0]
        self._check(lib().mpc_batch_track_init(self._h, x.shape[0], _p(x), T, v_ref), "mpc_batch_track_init")

    def track_run(self, n_ticks):
        B = self._track_B
        x, u = np.zeros((B, 3)), np.zeros((B, 2))
        target, status = np.zeros(B, np.int32), np.zeros(B, np.int32)
        self._check(lib().mpc_batch_track_run(self._h, B, n_ticks, _p(x), _p(u), _pi(target), _pi(status)), "mpc_batch_track_run")
        return dict(x=x, u=u, target=target, status=status)

    def get_diag(self, B=None):
        """Per-instance diagnostic bits of the last solve (MPC_DIAG_* in include/mpc_batch.h)."""
        B = self.capacity if B is None else B
        d = np.zeros(B, np.int32)
        self._check(lib().mpc_batch_get_diag(self._h, B, _pi(d)), "mpc_batch_get_diag")
        return d

    MODELS = {"auto": 0, "lane": 1, "wave": 2, "waveseq": 3, "wave1": 4, "wave2": 5, "wave2seq": 6}

    def set_model(self, model):
        """Execution model of the solve kernels: 'auto', 'lane', 'wave', 'waveseq' (include/mpc_batch.h)."""
        self._check(lib().mpc_batch_set_model(self._h, self.MODELS[model]), "mpc_batch_set_model")

    def reset(self, mask=None):
        if mask is None:
            rc = lib().mpc_batch_reset(self._h, None)
        else:
            m = np.ascontiguousarray(mask, np.uint8)
            if m.shape != (self.capacity,):
                raise ValueError("mask must have one entry per capsule")
            rc = lib().mpc_batch_reset(self._h, m.ctypes.data_as(ctypes.POINTER(ctypes.c_ubyte)))
        self._check(rc, "mpc_batch_reset")

    def shift(self, mask=None):
        """Shifted warm start: stage k of every (selected) capsule's iterate takes stage k+1's values."""
        if mask is None:
            rc = lib().mpc_batch_shift(self._h, None)
        else:
            m = np.ascontiguousarray(mask, np.uint8)
            if m.shape != (self.capacity,):
                raise ValueError("mask must have one entry per capsule")
            rc = lib().mpc_batch_shift(self._h, m.ctypes.data_as(ctypes.POINTER(ctypes.c_ubyte)))
        self._check(rc, "mpc_batch_shift")

    def track_set_shift(self, on):
        self._check(lib().mpc_batch_track_set_shift(self._h, 1 if on else 0), "mpc_batch_track_set_shift")

    def sim_step(self, x, u):
        x = np.ascontiguousarray(x, np.float64).copy()
        u = np.ascontiguousarray(u, np.float64)
        self._check(lib().mpc_batch_sim_step(self._h, x.shape[0], _p(x), _p(u)), "mpc_batch_sim_step")
        return x

    def debug_linearize(self, yref, yref_e):
        """QP data of the preparation phase at the capsules' current iterate (test hook):
        dict(ab [B,N+1,8], b [B,N+1,3], g [B,N+1,5], d [B,N+1,8])."""
        yref = np.ascontiguousarray(yref, np.float64)
        yref_e = np.ascontiguousarray(yref_e, np.float64)
        B = yref.shape[0]
        out = np.zeros((B, self.N + 1, 24))
        self._check(lib().mpc_batch_debug_linearize(self._h, B, _p(yref), _p(yref_e), _p(out)), "mpc_batch_debug_linearize")
        return dict(ab=out[:, :, 0:8], b=out[:, :, 8:11], g=out[:, :, 11:16], d=out[:, :, 16:24])

    def device_bytes(self):
        return int(lib().mpc_batch_device_bytes(self._h))

    def last_phase0_ms(self):
        return float(lib().mpc_batch_last_phase0_ms(self._h))

    def last_kernel_ms(self):
        return float(lib().mpc_batch_last_kernel_ms(self._h))

    def close(self):
        if self._h:
            lib().mpc_batch_free(self._h)
            self._h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
