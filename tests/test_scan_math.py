"""The parallel-in-time Riccati recursion of the wave kernel (mpc_amd/csrc/rti_wave.hpp,
factor_scan) restated in numpy: an associative suffix scan over conditional value-function
elements (A, C, J) [Sarkka & Garcia-Fernandez, IEEE TAC 2023] must reproduce the sequential
recursion P_k = Q_k + A^T (P^-1 + B R^-1 B^T)^-1 A on the IPM's own matrices — including the
barrier terms lam/t of a converged interior-point iterate (up to ~1e11) — to rounding."""
import numpy as np
import pytest

from mpc_amd import workload
from tests import oracle_lib


def _seq(A, B, Q, R, QN):
    N = len(A)
    P = [None] * (N + 1)
    P[N] = QN
    for k in range(N - 1, -1, -1):
        Pn = P[k + 1]
        K = -np.linalg.solve(R[k] + B[k].T @ Pn @ B[k], B[k].T @ Pn @ A[k])
        P[k] = Q[k] + A[k].T @ Pn @ A[k] + (A[k].T @ Pn @ B[k]) @ K
    return P


def _adj_inv(M):   # the kernel's 3x3 inverse: adjugate / determinant
    a, b, c, d, e, f, g, h, i = M.ravel()
    cof = np.array([[e * i - f * h, f * g - d * i, d * h - e * g],
                    [c * h - b * i, a * i - c * g, b * g - a * h],
                    [b * f - c * e, c * d - a * f, a * e - b * d]])
    return cof.T / (a * cof[0, 0] + b * cof[0, 1] + c * cof[0, 2])


def _combine(e1, e2):
    A1, C1, J1 = e1
    A2, C2, J2 = e2
    W = _adj_inv(np.eye(3) + C1 @ J2)
    G = A2 @ W
    return G @ A1, G @ C1 @ A2.T + C2, A1.T @ (J2 @ W) @ A1 + J1


def _scan(A, B, Q, R, QN):
    N = len(A)
    e = [(A[k], B[k] @ np.diag(1.0 / np.diag(R[k])) @ B[k].T, Q[k]) for k in range(N)] + [(np.zeros((3, 3)), np.zeros((3, 3)), QN)]
    d = 1
    while d < N + 1:   # Hillis-Steele: every "lane" k combines with lane k + d
        e = [_combine(e[k], e[k + d]) if k + d <= N else e[k] for k in range(N + 1)]
        d *= 2
    return [x[2] for x in e]


@pytest.mark.parametrize("name", ["mobile_robot", "park"])
def test_riccati_scan_equals_sequential_recursion(name):
    sp, wl = workload.make(name, 12)
    N = sp.N
    W, We = np.array(sp.W[:]), np.array(sp.W_e[:])
    worst, gmax = 0.0, 0.0
    x0, yref, yref_e = wl.tick(0)
    for i in range(12):
        d = oracle_lib.rti_debug(sp, x0[i], yref[i], yref_e[i])
        lam, t = d["lam"], d["t"]
        Q, R = [], []
        for k in range(N):
            nb = 2 if k == 0 else 4
            G = np.zeros(4)
            G[:nb] = lam[k, :nb] / t[k, :nb] + lam[k, 4:4 + nb] / t[k, 4:4 + nb]
            gmax = max(gmax, G.max())
            R.append(np.diag(sp.Ts * W[3:5] + 1e-15 + G[:2]))
            Q.append(np.diag(np.concatenate([sp.Ts * W[:2] + G[2:], [sp.Ts * W[2]]]) + 1e-15))
        QN = np.diag(We + 1e-15)
        Ps, Pc = _seq(d["A"], d["B"], Q, R, QN), _scan(d["A"], d["B"], Q, R, QN)
        worst = max(worst, max(np.abs(Ps[k] - Pc[k]).max() / np.abs(Ps[k]).max() for k in range(1, N + 1)))
    assert gmax > 1e3          # the barrier terms of a converged iterate are in play
    assert worst < 1e-13, worst
