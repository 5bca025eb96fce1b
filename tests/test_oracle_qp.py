"""Oracle pins G3 + G4: the QP the oracle builds and the solution its
HPIPM-style IPM returns, certified independently in numpy/scipy.  CPU only.

  * the linearisation data (g, d, b) is re-derived here from the iterate;
  * the IPM output must satisfy the QP's KKT conditions (a certificate for a convex QP);
  * the "polished" oracle (exit tolerances 1e-12) must agree to 1e-9 with scipy's exact
    bounded-variable least-squares active-set solver on the condensed problem;
  * the gap between HPIPM-BALANCE termination and the exact QP solution is measured
    (≈1e-4: SURVEY.md §7 hard part 1 — parity needs the same IPM path, not just the same QP);
  * repeated RTI steps converge to a point that satisfies the NLP's KKT conditions.
"""
import numpy as np
import pytest
from scipy.optimize import lsq_linear

from mpc_amd import variants, workload

VAR_ORDER_W = [3, 4, 0, 1, 2]  # z = [v, delta, x, y, psi] -> index into W = diag(x,y,psi,v,delta)


def _weights(sp):
    W = np.array(sp.W[:])
    H = sp.Ts * W[VAR_ORDER_W]
    He = np.array(sp.W_e[:])
    return H, He


def kkt_residuals(sp, dbg, yref, yref_e):
    """Max-norm KKT residuals of the full-space QP at the oracle's (dz, pi, lam, t)."""
    N = sp.N
    H, He = _weights(sp)
    A, B, b, g, d, dz, pi, lam = (dbg[k] for k in ("A", "B", "b", "g", "d", "dz", "pi", "lam"))
    r_dyn = r_stat = r_feas = r_comp = 0.0
    lb = np.array([sp.lbu[0], sp.lbu[1], sp.lbx[0], sp.lbx[1]])
    ub = np.array([sp.ubu[0], sp.ubu[1], sp.ubx[0], sp.ubx[1]])
    for k in range(N + 1):
        z = np.concatenate([dbg["u_in"][k] if k < N else np.zeros(2), dbg["x_in"][k]])
        # independent re-derivation of the linearisation
        if k < N:
            yr = np.concatenate([yref[k, 3:5], yref[k, 0:3]])
            g_here = H * (z - yr)
        else:
            g_here = np.concatenate([np.zeros(2), He * (z[2:] - yref_e)])
        nb = 0 if k == N else (2 if k == 0 else 4)
        if k == 0:
            np.testing.assert_allclose(g[k][:2], g_here[:2], rtol=1e-15, atol=1e-15)
        else:
            np.testing.assert_allclose(g[k], g_here, rtol=1e-15, atol=1e-15)
        np.testing.assert_allclose(d[k][:nb], (lb - z[:4])[:nb], rtol=0, atol=1e-15)
        np.testing.assert_allclose(d[k][4:4 + nb], (z[:4] - ub)[:nb], rtol=0, atol=1e-15)
        # stationarity
        Hk = H if k < N else np.concatenate([np.zeros(2), He])
        s = Hk * dz[k] + g_here
        if k > 0:
            s[2:] -= pi[k - 1]
        if k < N:
            s[:2] += B[k].T @ pi[k]
            s[2:] += A[k].T @ pi[k]
        s[:nb] += lam[k][4:4 + nb] - lam[k][:nb]
        rows = slice(0, 2) if k == 0 else (slice(2, 5) if k == N else slice(0, 5))
        r_stat = max(r_stat, np.abs(s[rows]).max())
        # feasibility + complementarity
        t_l = dz[k][:nb] - d[k][:nb]
        t_u = -dz[k][:nb] - d[k][4:4 + nb]
        if nb:
            r_feas = max(r_feas, max(0.0, -t_l.min(), -t_u.min()))
            assert (lam[k][:nb] >= 0).all() and (lam[k][4:4 + nb] >= 0).all()
            r_comp = max(r_comp, np.abs(lam[k][:nb] * t_l).max(), np.abs(lam[k][4:4 + nb] * t_u).max())
        if k < N:
            r_dyn = max(r_dyn, np.abs(dz[k + 1][2:] - (A[k] @ dz[k][2:] + B[k] @ dz[k][:2] + b[k])).max())
    return dict(dyn=r_dyn, stat=r_stat, feas=r_feas, comp=r_comp)


@pytest.mark.parametrize("name", ["mobile_robot", "mobile_robot_18", "mobile_robot_25", "park"])
def test_g3_kkt_certificate(oracle, name):
    sp, wl = workload.make(name, 6)
    x0, yref, yref_e = wl.tick(0)
    for i in range(6):
        xt = ut = None
        for rep in range(3):  # cold iterate, then two warm ones
            dbg = oracle.rti_debug(sp, x0[i], yref[i], yref_e[i], xt, ut)
            assert dbg["status"] == 0
            r = kkt_residuals(sp, dbg, yref[i], yref_e[i])
            # HPIPM BALANCE exit tolerances (res_g 1e-6, res_b/d/m 1e-8)
            assert r["dyn"] < 1e-8 and r["stat"] < 1e-6 and r["feas"] < 1e-8 and r["comp"] < 2e-8, r
            xt, ut = dbg["xtraj"], dbg["utraj"]


def _condensed_bvls(sp, dbg, yref, yref_e):
    """Exact solution of the QP by eliminating states and calling scipy's BVLS; valid when the
    position bounds are inactive (checked by the caller)."""
    N = sp.N
    H, He = _weights(sp)
    A, B, b = dbg["A"], dbg["B"], dbg["b"]
    nU = 2 * N
    # dx_k = Mx[k] du + mx[k]
    Mx = np.zeros((N + 1, 3, nU))
    mx = np.zeros((N + 1, 3))
    mx[0] = dbg["dz"][0][2:]
    for k in range(N):
        Mx[k + 1] = A[k] @ Mx[k]
        Mx[k + 1][:, 2 * k:2 * k + 2] += B[k]
        mx[k + 1] = A[k] @ mx[k] + b[k]
    rows, rhs = [], []
    for k in range(N + 1):
        if k < N:
            yr = np.concatenate([yref[k, 3:5], yref[k, 0:3]])
            z = np.concatenate([dbg["u_in"][k], dbg["x_in"][k]])
            w = np.sqrt(H)
            Eu = np.zeros((2, nU)); Eu[0, 2 * k] = 1; Eu[1, 2 * k + 1] = 1
            rows.append(w[:2, None] * Eu); rhs.append(-w[:2] * (z[:2] - yr[:2]))
            if k > 0:
                rows.append(w[2:, None] * Mx[k]); rhs.append(-w[2:] * (z[2:] + mx[k] - yr[2:]))
        else:
            w = np.sqrt(He)
            rows.append(w[:, None] * Mx[k]); rhs.append(-w * (dbg["x_in"][k] + mx[k] - yref_e))
    Amat, bvec = np.vstack(rows), np.concatenate(rhs)
    lo = np.tile([sp.lbu[0], sp.lbu[1]], N) - dbg["u_in"].ravel()
    hi = np.tile([sp.ubu[0], sp.ubu[1]], N) - dbg["u_in"].ravel()
    sol = lsq_linear(Amat, bvec, bounds=(lo, hi), method="bvls", tol=1e-15, max_iter=2000)
    du = sol.x
    dx = np.array([Mx[k] @ du + mx[k] for k in range(N + 1)])
    return du.reshape(N, 2), dx


@pytest.mark.parametrize("name,N", [("mobile_robot", 12), ("mobile_robot", 50), ("park", 20), ("mobile_robot_25", 25)])
def test_g3_independent_solver_and_termination_gap(oracle, name, N):
    sp, wl = workload.make(name, 4, N=N)
    x0, yref, yref_e = wl.tick(0)
    worst_gap = 0.0
    for i in range(4):
        warm = oracle.rti_debug(sp, x0[i], yref[i], yref_e[i])  # one step to get a non-trivial iterate
        base = oracle.rti_debug(sp, x0[i], yref[i], yref_e[i], warm["xtraj"], warm["utraj"])
        oracle.set_ipm_override(1e-12, 1e-12, 1e-12, 1e-13, 100)
        try:
            pol = oracle.rti_debug(sp, x0[i], yref[i], yref_e[i], warm["xtraj"], warm["utraj"])
        finally:
            oracle.set_ipm_override()
        du, dx = _condensed_bvls(sp, pol, yref[i], yref_e[i])
        xnew = pol["x_in"] + dx
        lbx, ubx = np.array(sp.lbx[:]), np.array(sp.ubx[:])
        assert (xnew[1:N, :2] > lbx + 1e-6).all() and (xnew[1:N, :2] < ubx - 1e-6).all(), "position bound active: BVLS cross-check not applicable"
        np.testing.assert_allclose(pol["dz"][:N, :2], du, rtol=0, atol=1e-9)
        np.testing.assert_allclose(pol["dz"][:, 2:], dx, rtol=0, atol=1e-9)
        gap = np.abs(base["dz"] - pol["dz"]).max()
        worst_gap = max(worst_gap, gap)
    # HPIPM-BALANCE termination (res_g <= 1e-6 against a smallest Hessian entry Ts*W_delta =
    # 5e-3) stops up to ~2e-4 short of the exact QP solution (SURVEY.md §7 hard part 1,
    # measured 1.7e-4 here).  So 1e-6 parity with acados REQUIRES following the same IPM
    # path to the same exit iteration — which is why the HIP kernels restate the IPM step
    # for step instead of solving the QP "exactly".  This bound is a sanity check only.
    assert worst_gap < 1e-3, worst_gap


def test_g4_rti_fixed_point_is_nlp_kkt_point(oracle):
    sp, wl = workload.make("mobile_robot", 3)
    x0, yref, yref_e = wl.tick(0)
    for i in range(3):
        # with BALANCE exit tolerances every QP is ~1e-4 inexact, so the iteration settles
        # into a ~1e-3 neighbourhood; with polished QPs it converges to the NLP KKT point
        xt = ut = None
        for _ in range(25):
            dbg = oracle.rti_debug(sp, x0[i], yref[i], yref_e[i], xt, ut)
            assert dbg["status"] == 0
            xt, ut = dbg["xtraj"], dbg["utraj"]
        assert np.abs(oracle.rti_debug(sp, x0[i], yref[i], yref_e[i], xt, ut)["dz"]).max() < 5e-3
        oracle.set_ipm_override(1e-12, 1e-12, 1e-12, 1e-13, 100)
        try:
            for _ in range(150):  # Gauss-Newton: linear convergence
                dbg = oracle.rti_debug(sp, x0[i], yref[i], yref_e[i], xt, ut)
                xt, ut = dbg["xtraj"], dbg["utraj"]
            last = oracle.rti_debug(sp, x0[i], yref[i], yref_e[i], xt, ut)
        finally:
            oracle.set_ipm_override()
        assert np.abs(last["dz"]).max() < 1e-7          # no further step: SQP fixed point
        assert np.abs(last["b"]).max() < 1e-9           # multiple-shooting continuity satisfied
        np.testing.assert_allclose(last["x_in"][0], x0[i], rtol=0, atol=1e-14)  # x0 embedding
        r = kkt_residuals(sp, last, yref[i], yref_e[i])
        assert r["stat"] < 1e-9 and r["comp"] < 1e-9


def test_failure_keeps_iterate(oracle):
    # iter_max = 1 cannot converge: HPIPM reports MAX_ITER, which SQP_RTI accepts (status 0);
    # a NaN input makes the QP fail: status 4 and the iterate and u0 stay untouched
    sp, wl = workload.make("mobile_robot", 2)
    x0, yref, yref_e = wl.tick(0)
    sp1 = sp.copy(qp_iter_max=1)
    r = oracle.rti_batch(sp1, x0, yref, yref_e)
    assert (r["status"] == 0).all() and (r["qp_iter"] == 1).all()
    bad = yref.copy(); bad[1, 3, 0] = np.nan
    xt = np.full((2, sp.N + 1, 3), 0.25); ut = np.full((2, sp.N, 2), 0.125)
    r = oracle.rti_batch(sp, x0, bad, yref_e, xt, ut)
    assert r["status"][0] == 0 and r["status"][1] == variants.QP_FAILURE
    assert np.array_equal(r["xtraj"][1], xt[1]) and np.array_equal(r["utraj"][1], ut[1])
    assert np.array_equal(r["u0"][1], ut[1, 0])


def test_iteration_statistics(oracle):
    # regression on the IPM path itself: BALANCE mode on the headline workload
    sp, wl = workload.make("mobile_robot", 256)
    x0, yref, yref_e = wl.tick(0)
    r = oracle.rti_batch(sp, x0, yref, yref_e, want_stats=True)
    assert (r["status"] == 0).all()
    assert 5 <= r["qp_iter"].min() and r["qp_iter"].max() <= 50
    st = r["stats"]
    assert st[:, 2].sum() == 0          # the lq_fact=1 residual test never asks for LQ here
    assert st[:, 5].max() < 1e-9        # linear-system residual after Cholesky Riccati


def test_lq_factorisation_equals_cholesky_and_fires_on_heavy_weights(oracle):
    """HPIPM's LQ variant of the factorisation (lq_fact): (i) forced on every iteration (lq_fact = 2)
    it reproduces the Cholesky path (lq_fact = 0) to rounding on the regular workloads — same
    iteration counts, 1e-12; (ii) with the mode's lq_fact = 1 it fires on a heavy-weight, bounds-active
    workload (the residual test of OCP_QP_IPM_DELTA_STEP), every QP still exits with status 0 and its
    final residuals below the exit tolerances."""
    from mpc_amd import workload
    try:
        for name in ("mobile_robot", "park"):
            sp, wl = workload.make(name, 64)
            x0, yref, yref_e = wl.tick(0)
            out = {}
            for lq in (0, 2):
                oracle.set_lq_fact(lq)
                r0 = oracle.rti_batch(sp, x0, yref, yref_e)
                out[lq] = oracle.rti_batch(sp, *wl.tick(1), r0["xtraj"], r0["utraj"])
            assert np.array_equal(out[0]["qp_iter"], out[2]["qp_iter"])
            assert np.abs(out[0]["utraj"] - out[2]["utraj"]).max() < 1e-12 and np.abs(out[0]["xtraj"] - out[2]["xtraj"]).max() < 1e-12
    finally:
        oracle.set_lq_fact(-1)
    B = 256
    sp, wl = workload.make("mobile_robot", B)
    sp = sp.copy(lbx=(-0.3, -0.3), ubx=(0.3, 0.3), W=tuple(1e6 * w for w in sp.W), W_e=tuple(1e6 * w for w in sp.W_e))
    x0, yref, yref_e = (a.copy() for a in wl.tick(0))
    shift = x0[:, :2].copy()
    x0[:, :2] -= shift; yref[:, :, :2] -= shift[:, None, :]; yref_e[:, :2] -= shift
    r = oracle.rti_batch(sp, x0, yref, yref_e, want_stats=True)
    st = r["stats"]
    assert (st[:, 2] > 0).sum() >= 5 and (st[:, 5] > 1e-5).sum() == (st[:, 2] > 0).sum()   # fired exactly where the residual asked for it
    assert (r["status"] == 0).all() and (st[:, 1] == 0).all()
