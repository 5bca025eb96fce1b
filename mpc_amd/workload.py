"""Deterministic synthetic OCP instances — numpy PCG64, seed 20241110.

Each instance follows an arc reference of the kind scripts/path2.py produces
(waypoint spacing v_ref*Ts, scripts/path2.py:253): start pose uniform over the map
(≈20 m x 14 m, scripts/paths/state_refs_path118.txt), constant curvature
kappa ~ U(-1.5, 1.5) 1/m, reference speed = the variant's cruise speed, steering
reference forced to 0 and terminal reference = the FIRST waypoint of the window, both
as Optimizer::update_and_solve does (src/optimizer.cpp:244-251, 271).
The measured state is the reference pose plus N(0, (0.05 m, 0.05 m, 0.1 rad)).
"""
import numpy as np

from . import variants

SEED = 20241110
SIGMA = (0.05, 0.05, 0.1)


def _arc(x0, y0, psi0, kappa, s):
    """Pose at arc length s (arrays broadcast) on a constant-curvature arc."""
    small = np.abs(kappa) < 1e-9
    k = np.where(small, 1.0, kappa)
    psi = psi0 + kappa * s
    x = np.where(small, x0 + s * np.cos(psi0), x0 + (np.sin(psi) - np.sin(psi0)) / k)
    y = np.where(small, y0 + s * np.sin(psi0), y0 - (np.cos(psi) - np.cos(psi0)) / k)
    return x, y, psi


class ArcTracking:
    """A batch of B arc-tracking instances; ``tick(t)`` returns the inputs of control
    tick t: (x0 [B,3], yref [B,N,5], yref_e [B,3]) in the C-ABI's AoS layouts."""

    def __init__(self, spec, B, v_ref, seed=SEED, first=0):
        # `first` offsets the instance index so that ranks of a sharded run draw disjoint,
        # reproducible slices of one global batch.
        self.spec, self.B, self.v_ref = spec, int(B), float(v_ref)
        self.N, self.Ts = int(spec.N), float(spec.Ts)
        rng = np.random.Generator(np.random.PCG64([seed, 0]))
        total = first + self.B
        u = rng.random((total, 4))[first:]
        self.px = 20.0 * u[:, 0]
        self.py = 14.0 * u[:, 1]
        self.psi = -np.pi + 2.0 * np.pi * u[:, 2]
        self.kappa = -1.5 + 3.0 * u[:, 3]
        self.seed, self.first = seed, first

    def tick(self, t=0):
        B, N = self.B, self.N
        ds = self.v_ref * self.Ts
        s = (t + np.arange(N + 1))[None, :] * ds
        x, y, psi = _arc(self.px[:, None], self.py[:, None], self.psi[:, None], self.kappa[:, None], s)
        yref = np.zeros((B, N, 5))
        yref[:, :, 0] = x[:, :N]
        yref[:, :, 1] = y[:, :N]
        yref[:, :, 2] = psi[:, :N]
        yref[:, :, 3] = self.v_ref
        yref[:, :, 4] = 0.0
        yref_e = np.stack([x[:, 0], y[:, 0], psi[:, 0]], axis=1)  # reference quirk: first waypoint
        rng = np.random.Generator(np.random.PCG64([self.seed, 1 + t]))
        noise = rng.standard_normal((self.first + B, 3))[self.first:] * np.asarray(SIGMA)
        x0 = yref_e + noise
        return np.ascontiguousarray(x0), np.ascontiguousarray(yref), np.ascontiguousarray(yref_e)


class Parking:
    """Config 4: constant target (0.63, 0.32, pi) in the park frame, x0 ~ U(±0.8, ±0.5, ±0.5)
    around it (include/optimizer.hpp:167-176, 302): bounds on v and delta become active."""

    TARGET = (0.63, 0.32, np.pi)

    def __init__(self, spec, B, seed=SEED, first=0):
        self.spec, self.B, self.N = spec, int(B), int(spec.N)
        rng = np.random.Generator(np.random.PCG64([seed, 4]))
        u = rng.random((first + self.B, 3))[first:]
        half = np.array([0.8, 0.5, 0.5])
        self.x0 = np.asarray(self.TARGET) + (2.0 * u - 1.0) * half

    def tick(self, t=0):
        B, N = self.B, self.N
        yref = np.zeros((B, N, 5))
        yref[:, :, 0:3] = np.asarray(self.TARGET)
        yref_e = np.tile(np.asarray(self.TARGET), (B, 1))
        return np.ascontiguousarray(self.x0), yref, np.ascontiguousarray(yref_e)


def make(name, B, seed=SEED, first=0, **spec_overrides):
    """(spec, workload) for a variant name of mpc_amd.variants."""
    sp = variants.spec(name, **spec_overrides)
    if name == "park":
        return sp, Parking(sp, B, seed, first)
    return sp, ArcTracking(sp, B, variants.cruise_speed(name), seed, first)
