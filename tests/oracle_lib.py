"""ctypes binding of the CPU oracle (oracle/_build/libmpc_oracle.so) and of the
reference's own CasADi model functions compiled into oracle/_ref/libref_model.so.

TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py — never by the product package mpc_amd.
"""
import ctypes
import os
import subprocess

import numpy as np

from mpc_amd.variants import OcpSpec

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "_build", "libmpc_oracle.so")
REF_SO = os.path.join(ORACLE_DIR, "_ref", "libref_model.so")

_dp = ctypes.POINTER(ctypes.c_double)
_ip = ctypes.POINTER(ctypes.c_int)


def _p(a):
    return None if a is None else a.ctypes.data_as(_dp)


def _pi(a):
    return None if a is None else a.ctypes.data_as(_ip)


def build():
    """Compile the oracle (and oracle/_ref when /root/reference is mounted)."""
    subprocess.run(["make", "-C", ORACLE_DIR, "-s"], check=True)


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(ORACLE_SO):
            build()
        L = ctypes.CDLL(ORACLE_SO)
        L.oracle_vde_forw.argtypes = [ctypes.c_double, _dp, _dp, _dp, _dp, _dp]
        L.oracle_ode_fun.argtypes = [ctypes.c_double, _dp, _dp, _dp]
        L.oracle_erk4.argtypes = [ctypes.c_double, ctypes.c_double, _dp, _dp, _dp, _dp, _dp]
        L.oracle_sim_step.argtypes = [ctypes.POINTER(OcpSpec), ctypes.c_int, _dp, _dp]
        L.oracle_rti_batch.argtypes = [ctypes.POINTER(OcpSpec), ctypes.c_int, _dp, _dp, _dp, _dp, _dp, _dp,
                                       _dp, _ip, _ip, _dp, ctypes.c_int]
        L.oracle_rti_batch.restype = ctypes.c_int
        L.oracle_sqp_batch.argtypes = [ctypes.POINTER(OcpSpec), ctypes.c_int] + [_dp] * 8 + [_ip, _ip, _ip, _dp, ctypes.c_int]
        L.oracle_sqp_batch.restype = ctypes.c_int
        L.oracle_rti_debug.argtypes = [ctypes.POINTER(OcpSpec)] + [_dp] * 14 + [_ip, _dp]
        L.oracle_rti_debug.restype = ctypes.c_int
        L.oracle_count_ops.argtypes = [ctypes.POINTER(OcpSpec), _dp, _dp, _dp, _dp, _dp, _dp, _ip]
        L.oracle_count_ops.restype = ctypes.c_int
        L.oracle_max_threads.restype = ctypes.c_int
        L.oracle_set_ipm_override.argtypes = [ctypes.c_double] * 4 + [ctypes.c_int]
        _lib = L
    return _lib


def set_ipm_override(res_g=0.0, res_b=0.0, res_d=0.0, res_m=0.0, iter_max=0):
    """Polish mode: override the IPM exit tolerances (0 = mode default)."""
    lib().oracle_set_ipm_override(res_g, res_b, res_d, res_m, iter_max)


def set_lq_fact(lq_fact=-1):
    """HPIPM's lq_fact in the oracle: 0 Cholesky only, 1 switch to LQ when the residual asks for it
    (the mode's value), 2 LQ always; -1 restores the mode's value."""
    lib().oracle_set_lq_fact(int(lq_fact))


def vde_forw(x, Sx, Su, u, L=0.0):
    out = np.zeros(18)
    lib().oracle_vde_forw(L, _p(np.ascontiguousarray(x, float)), _p(np.ascontiguousarray(Sx, float)),
                          _p(np.ascontiguousarray(Su, float)), _p(np.ascontiguousarray(u, float)), _p(out))
    return out


def ode_fun(x, u, L=0.0):
    out = np.zeros(3)
    lib().oracle_ode_fun(L, _p(np.ascontiguousarray(x, float)), _p(np.ascontiguousarray(u, float)), _p(out))
    return out


def erk4(x, u, h, L=0.0):
    xn, A, B = np.zeros(3), np.zeros(9), np.zeros(6)
    lib().oracle_erk4(L, h, _p(np.ascontiguousarray(x, float)), _p(np.ascontiguousarray(u, float)), _p(xn), _p(A), _p(B))
    return xn, A.reshape(3, 3).T.copy(), B.reshape(2, 3).T.copy()  # col-major -> numpy matrices


def sim_step(spec, x, u):
    x = np.ascontiguousarray(x, float).copy()
    u = np.ascontiguousarray(u, float)
    lib().oracle_sim_step(ctypes.byref(spec), x.shape[0], _p(x), _p(u))
    return x


def rti_batch(spec, x0, yref, yref_e, xtraj=None, utraj=None, nthreads=0, want_stats=False):
    """One RTI step per instance.  Returns dict(u0, xtraj, utraj, pi, status, qp_iter[, stats])."""
    B, N = x0.shape[0], spec.N
    x0 = np.ascontiguousarray(x0, float)
    yref = np.ascontiguousarray(yref, float)
    yref_e = np.ascontiguousarray(yref_e, float)
    assert yref.shape == (B, N, 5) and yref_e.shape == (B, 3)
    xt = np.zeros((B, N + 1, 3)) if xtraj is None else np.ascontiguousarray(xtraj, float).copy()
    ut = np.zeros((B, N, 2)) if utraj is None else np.ascontiguousarray(utraj, float).copy()
    pi = np.zeros((B, N, 3))
    u0 = np.zeros((B, 2))
    status = np.zeros(B, np.int32)
    it = np.zeros(B, np.int32)
    stats = np.zeros((B, 8)) if want_stats else None
    rc = lib().oracle_rti_batch(ctypes.byref(spec), B, _p(x0), _p(yref), _p(yref_e), _p(xt), _p(ut), _p(pi),
                                _p(u0), _pi(status), _pi(it), _p(stats), nthreads)
    assert rc == 0
    out = dict(u0=u0, xtraj=xt, utraj=ut, pi=pi, status=status, qp_iter=it)
    if want_stats:
        out["stats"] = stats
    return out


def sqp_batch(spec, x0, yref, yref_e, xtraj=None, utraj=None, pi=None, lam=None, nthreads=0):
    """Full SQP per instance (spec.sqp_max_iter, spec.nlp_tol).  Returns dict(u0, xtraj, utraj, pi, lam,
    status, sqp_iter, qp_iter, res [B,4] = stat, eq, ineq, comp)."""
    B, N = x0.shape[0], spec.N
    x0 = np.ascontiguousarray(x0, float)
    yref = np.ascontiguousarray(yref, float)
    yref_e = np.ascontiguousarray(yref_e, float)
    xt = np.zeros((B, N + 1, 3)) if xtraj is None else np.ascontiguousarray(xtraj, float).copy()
    ut = np.zeros((B, N, 2)) if utraj is None else np.ascontiguousarray(utraj, float).copy()
    pp = np.zeros((B, N, 3)) if pi is None else np.ascontiguousarray(pi, float).copy()
    ll = np.zeros((B, N + 1, 8)) if lam is None else np.ascontiguousarray(lam, float).copy()
    u0 = np.zeros((B, 2))
    status, sit, qit = np.zeros(B, np.int32), np.zeros(B, np.int32), np.zeros(B, np.int32)
    res = np.zeros((B, 4))
    rc = lib().oracle_sqp_batch(ctypes.byref(spec), B, _p(x0), _p(yref), _p(yref_e), _p(xt), _p(ut), _p(pp), _p(ll),
                                _p(u0), _pi(status), _pi(sit), _pi(qit), _p(res), nthreads)
    assert rc == 0
    return dict(u0=u0, xtraj=xt, utraj=ut, pi=pp, lam=ll, status=status, sqp_iter=sit, qp_iter=qit, res=res)


def rti_debug(spec, x0, yref, yref_e, xtraj=None, utraj=None):
    """Single instance; returns the QP (A, B, b, g, d) and its solution (dz, pi, lam, t)."""
    N = spec.N
    xt = np.zeros((N + 1, 3)) if xtraj is None else np.ascontiguousarray(xtraj, float).copy()
    ut = np.zeros((N, 2)) if utraj is None else np.ascontiguousarray(utraj, float).copy()
    x_in, u_in = xt.copy(), ut.copy()
    A, Bm, b = np.zeros((N, 9)), np.zeros((N, 6)), np.zeros((N, 3))
    g, d = np.zeros((N + 1, 5)), np.zeros((N + 1, 8))
    dz, qpi = np.zeros((N + 1, 5)), np.zeros((N, 3))
    lam, t = np.zeros((N + 1, 8)), np.zeros((N + 1, 8))
    it = np.zeros(1, np.int32)
    res = np.zeros(4)
    st = lib().oracle_rti_debug(ctypes.byref(spec), _p(np.ascontiguousarray(x0, float)),
                                _p(np.ascontiguousarray(yref, float)), _p(np.ascontiguousarray(yref_e, float)),
                                _p(xt), _p(ut), _p(A), _p(Bm), _p(b), _p(g), _p(d), _p(dz), _p(qpi), _p(lam), _p(t),
                                _pi(it), _p(res))
    return dict(status=st, qp_iter=int(it[0]), res=res, x_in=x_in, u_in=u_in, xtraj=xt, utraj=ut,
                A=A.reshape(N, 3, 3).transpose(0, 2, 1).copy(), B=Bm.reshape(N, 2, 3).transpose(0, 2, 1).copy(),
                b=b, g=g, d=d, dz=dz, pi=qpi, lam=lam, t=t)


def count_ops(spec, x0, yref, yref_e, xtraj, utraj):
    out = np.zeros(9)
    it = np.zeros(1, np.int32)
    st = lib().oracle_count_ops(ctypes.byref(spec), _p(np.ascontiguousarray(x0, float)),
                                _p(np.ascontiguousarray(yref, float)), _p(np.ascontiguousarray(yref_e, float)),
                                _p(np.ascontiguousarray(xtraj, float)), _p(np.ascontiguousarray(utraj, float)),
                                _p(out), _pi(it))
    return dict(status=st, qp_iter=int(it[0]), add=out[0], mul=out[1], div=out[2], sqrt=out[3], trig=out[4], cmp=out[5],
                s_add=out[6], s_mul=out[7], s_div=out[8])


# ---- the reference's own CasADi functions (oracle/_ref), for golden-vector generation ----
_ref = None


def ref_lib():
    global _ref
    if _ref is None:
        _ref = ctypes.CDLL(REF_SO)
    return _ref


def ref_call(fname, args, out_sizes):
    """Call a CasADi-generated function `int f(const double** arg, double** res, int* iw, double* w, int mem)`."""
    f = getattr(ref_lib(), fname)
    f.restype = ctypes.c_int
    arr_in = [np.ascontiguousarray(a, float) for a in args]
    arr_out = [np.zeros(n) for n in out_sizes]
    ArgT = _dp * len(arr_in)
    ResT = _dp * len(arr_out)
    argp = ArgT(*[a.ctypes.data_as(_dp) if a.size else ctypes.cast(None, _dp) for a in arr_in])
    resp = ResT(*[a.ctypes.data_as(_dp) for a in arr_out])
    iw = (ctypes.c_int * 8)()
    w = (ctypes.c_double * 8)()
    rc = f(argp, resp, iw, w, 0)
    assert rc == 0
    return arr_out
