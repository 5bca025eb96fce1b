"""Oracle-backed stand-in with the BatchSolver interface (solve / sim_step / reset): lets the
CPU suite exercise the host-side `Optimizer` mirror, and serves as the checker for the GPU
closed-loop tests.  TEST INFRASTRUCTURE ONLY — never imported by mpc_amd."""
import numpy as np

from tests import oracle_lib


class OracleSolver:
    def __init__(self, spec, capacity=1):
        self.spec, self.N, self.capacity = spec, int(spec.N), capacity
        self.reset()

    def reset(self, mask=None):
        self.xtraj = np.zeros((self.capacity, self.N + 1, 3))
        self.utraj = np.zeros((self.capacity, self.N, 2))
        self.u0_prev = np.zeros((self.capacity, 2))

    def solve(self, x0, yref, yref_e, xtraj=None, utraj=None):
        B = x0.shape[0]
        xt = self.xtraj[:B] if xtraj is None else xtraj
        ut = self.utraj[:B] if utraj is None else utraj
        r = oracle_lib.rti_batch(self.spec, x0, yref, yref_e, xt, ut)
        self.xtraj[:B], self.utraj[:B] = r["xtraj"], r["utraj"]
        return r

    def shift(self, mask=None):
        sel = slice(None) if mask is None else np.asarray(mask, bool)
        self.xtraj[sel, :-1] = self.xtraj[sel, 1:]
        self.utraj[sel, :-1] = self.utraj[sel, 1:]

    def sim_step(self, x, u):
        return oracle_lib.sim_step(self.spec, x, u)

    def close(self):
        pass
