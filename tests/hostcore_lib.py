"""ctypes binding of tests/hostbuild/_build/libhostcore.so: the product's per-lane algorithm
(mpc_amd/csrc/rti_core.hpp) compiled for the host.  TEST INFRASTRUCTURE — lets the CPU suite
check the structure-exploiting kernel algorithm against the dense oracle without a GPU."""
import ctypes
import os
import subprocess

import numpy as np

from mpc_amd.variants import OcpSpec

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "hostbuild", "_build", "libhostcore.so")
_dp = ctypes.POINTER(ctypes.c_double)
_ip = ctypes.POINTER(ctypes.c_int)
_lib = None


def lib():
    global _lib
    if _lib is None:
        subprocess.run(["make", "-C", os.path.join(HERE, "hostbuild"), "-s"], check=True)
        _lib = ctypes.CDLL(SO)
        _lib.hostcore_rti_batch.argtypes = [ctypes.POINTER(OcpSpec), ctypes.c_int] + [_dp] * 7 + [_ip] * 3 + [_dp, _ip, _dp]
    return _lib


def rti_batch(spec, x0, yref, yref_e, xtraj=None, utraj=None, pi=None, lam=None):
    """One RTI step per instance, or the full SQP loop when spec.sqp_max_iter > 1."""
    B, N = x0.shape[0], spec.N
    x0 = np.ascontiguousarray(x0, float); yref = np.ascontiguousarray(yref, float)
    yref_e = np.ascontiguousarray(yref_e, float)
    xt = np.zeros((B, N + 1, 3)) if xtraj is None else np.ascontiguousarray(xtraj, float).copy()
    ut = np.zeros((B, N, 2)) if utraj is None else np.ascontiguousarray(utraj, float).copy()
    pi = np.zeros((B, N, 3)) if pi is None else np.ascontiguousarray(pi, float).copy()
    lam = np.zeros((B, N + 1, 8)) if lam is None else np.ascontiguousarray(lam, float).copy()
    u0 = np.zeros((B, 2)); sit = np.zeros(B, np.int32); res = np.zeros((B, 4))
    st = np.zeros(B, np.int32); it = np.zeros(B, np.int32); dg = np.zeros(B, np.int32)
    P = lambda a: a.ctypes.data_as(_dp)
    I = lambda a: a.ctypes.data_as(_ip)
    rc = lib().hostcore_rti_batch(ctypes.byref(spec), B, P(x0), P(yref), P(yref_e), P(xt), P(ut), P(pi), P(u0),
                                  I(st), I(it), I(dg), P(lam), I(sit), P(res))
    assert rc == 0
    return dict(u0=u0, xtraj=xt, utraj=ut, pi=pi, lam=lam, status=st, qp_iter=it, diag=dg, sqp_iter=sit, res=res)
