"""Producer of `state_refs` / `input_refs`: the numerical part of the
reference's `Path` class (scripts/path2.py:14-99, 253-434) — spline resampling of planned runs
at density 1/(v_ref*T), duplicate filtering, smoothed curvature / heading / normals, speed ramp,
padding by the horizon.  The graph planner that produces the runs (GlobalPlanner, rospy, the
map yaml) is control plane and stays out of scope: runs are an input here.

Pinned by the reference's own data files (tests/golden/pathgen): its saved runs and the
state_refs / input_refs / kappa it generated from them."""
import numpy as np
from scipy.interpolate import UnivariateSpline, splev, splprep


def smooth_yaw_angles(yaw_angles):
    """scripts/path2.py:14-31: unwrap jumps larger than pi."""
    diffs = np.diff(yaw_angles)
    diffs[diffs > np.pi] -= 2 * np.pi
    diffs[diffs < -np.pi] += 2 * np.pi
    return np.concatenate(([yaw_angles[0]], yaw_angles[0] + np.cumsum(diffs)))


def compute_smooth_curvature(waypoints_x, waypoints_y, smooth_factor=0.1):
    """scripts/path2.py:33-55: cubic smoothing splines x(t), y(t) -> curvature, tangent angle, normals."""
    t = np.linspace(0, 1, len(waypoints_x))
    sx = UnivariateSpline(t, waypoints_x, k=3, s=smooth_factor)
    sy = UnivariateSpline(t, waypoints_y, k=3, s=smooth_factor)
    dx, dy = sx.derivative()(t), sy.derivative()(t)
    ddx, ddy = sx.derivative(n=2)(t), sy.derivative(n=2)(t)
    curvature = np.abs(dx * ddy - dy * ddx) / (dx ** 2 + dy ** 2) ** 1.5
    tangent = np.arctan2(dy, dx)
    normals = np.vstack((np.cos(tangent + np.pi / 2), np.sin(tangent + np.pi / 2))).T
    return curvature, tangent, normals


def filter_waypoints(waypoints, threshold=0.01, attributes=None):
    """scripts/path2.py:57-80: drop a waypoint closer than `threshold` to its predecessor IN THE
    ORIGINAL sequence (not to the last kept one)."""
    keep = [0] + [i for i in range(1, len(waypoints)) if np.linalg.norm(waypoints[i] - waypoints[i - 1]) >= threshold]
    w = np.asarray(waypoints)[keep]
    return (w, np.asarray(attributes)[keep]) if attributes is not None else w


def interpolate_waypoints(waypoints, num_points):
    """scripts/path2.py:82-92: interpolating parametric spline, num_points equally spaced in u."""
    tck, _ = splprep([waypoints[:, 0], waypoints[:, 1]], s=0)
    x, y = splev(np.linspace(0, 1, num_points), tck)
    return np.vstack((x, y)).T


def resample_run(run, v_ref, T):
    """One planned run [n,2] at density 1/(|v_ref|*T) waypoints per metre (scripts/path2.py:253-268)."""
    length = np.sum(np.linalg.norm(run[1:] - run[:-1], axis=1))
    density = 1.0 / abs(v_ref) / T
    return interpolate_waypoints(run, int(np.ceil(length * density)))


def references(runs, v_ref, N, T, resample=False):
    """runs: list of [n_i, 2] arrays (already at path density unless resample=True).
    Returns dict(state_refs [M,3], input_refs [M+1,2], kappa [M+1], normals [M+1,2], density) as
    scripts/path2.py:372-415 builds them (speedrun branch: no cross-walk / highway segments)."""
    if resample:
        runs = [resample_run(np.asarray(r, float), v_ref, T) for r in runs]
    wp = filter_waypoints(np.vstack(runs), 0.01)
    x, y = wp[:, 0].copy(), wp[:, 1].copy()
    density = 1.0 / abs(v_ref) / T
    kappa, theta, normals = compute_smooth_curvature(x, y)
    v_refs = v_ref / (1 + np.abs(kappa)) * (1 + np.abs(kappa))      # "linear speed profile" (:383)
    ramp = min(int(1 * density), len(v_refs))                        # 1 m ramp (:398-401)
    v_refs[:ramp] = np.linspace(0, v_ref, ramp)
    v_refs[-2:] = 0                                                  # stop at the end
    steer = 0 * kappa                                                # k_steer = 0 (:405-406)
    x = np.pad(x, (0, N + 4), "edge"); y = np.pad(y, (0, N + 4), "edge")
    kappa = np.pad(kappa, (0, N + 5), "edge"); theta = np.pad(theta, (0, N + 5), "edge")
    normals = np.pad(normals, ((0, N + 5), (0, 0)), "edge")
    v_refs = np.pad(v_refs, (0, N + 5), "edge"); steer = np.pad(steer, (0, N + 5), "edge")
    state_refs = np.vstack((x, y, theta[1:])).T                      # heading of the NEXT sample (:414)
    state_refs[:, 2] = smooth_yaw_angles(state_refs[:, 2])
    return dict(state_refs=state_refs, input_refs=np.vstack((v_refs, steer)).T, kappa=kappa, normals=normals,
                density=density)


def save_txt(filename, array, fmt="%.8f"):
    """The `%.8f` text format of scripts/paths/*.txt (np.savetxt, scripts/mpc_acados.py:248-254)
    that Optimizer::loadTxt reads; counterpart of Optimizer::saveToFile (src/optimizer.cpp:532-543)."""
    np.savetxt(filename, np.atleast_2d(array) if np.ndim(array) > 1 else np.asarray(array), fmt=fmt)
