"""Host-side `Optimizer` mirror (mpc_amd/optimizer.py) against the behaviours of the reference's
C++ class that SURVEY.md §3.5 lists (src/optimizer.cpp:208-455, include/optimizer.hpp:140-267).
The solver behind it is the CPU oracle stand-in (tests/oracle_solver.py) — these tests check
the host logic, not the HIP path; tests/test_gpu_parity.py::test_optimizer_closed_loop_gpu runs
the same loop on the GPU library."""
import math
import os

import numpy as np
import pytest

from mpc_amd.optimizer import Optimizer, load_txt, variant_for_speed
from tests.oracle_solver import OracleSolver

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def path150():
    return (load_txt(os.path.join(GOLD, "state_refs_path150_head.txt")),
            load_txt(os.path.join(GOLD, "input_refs_path150_head.txt")))


def make(v_ref=0.5, T=0.1, start=None, path=None):
    s, u = path or path150()
    x, y, yaw = start if start is not None else s[0]
    return Optimizer(T, 40, v_ref, x, y, yaw, state_refs=s, input_refs=u, solver_factory=OracleSolver)


def test_variant_selection():
    assert variant_for_speed(50) == "mobile_robot"
    assert variant_for_speed(30) == "mobile_robot"
    assert variant_for_speed(25) == "mobile_robot_25"
    assert variant_for_speed(18) == "mobile_robot_18"
    with pytest.raises(ValueError):
        variant_for_speed(20)
    o = make()
    assert (o.N, o.N_park) == (50, 49)      # dims come from the solver, not the ctor argument
    assert o.v_ref == 0.35                  # src/optimizer.cpp:174
    assert o.density == pytest.approx(1.0 / 0.1 / 0.5)


def test_load_txt(tmp_path):
    p = tmp_path / "a.txt"
    p.write_text("1 2 3\n4 5 6\n")
    assert np.array_equal(load_txt(str(p)), [[1, 2, 3], [4, 5, 6]])
    p.write_text("1 2 3\n4 5\n")
    with pytest.raises(RuntimeError):
        load_txt(str(p))


def test_closest_waypoint_window_and_ties():
    s = np.zeros((100, 3)); s[:, 0] = np.arange(100) * 0.1
    u = np.zeros((100, 2))
    o = make(path=(s, u), start=(0.25, 0.0, 0.0))
    # equidistant from waypoints 2 (0.2) and 3 (0.3): the downward scan with strict '<' keeps 3
    o.x_current[:] = (0.25, 0.0, 0.0)
    assert o.find_closest_waypoint(0, 99) == 3
    # default window: [last_waypoint_index, target + limit], limit = floor(3.95/(0.35*0.1)) = 112
    assert o._limit_value() == 112
    o.last_waypoint_index, o.target_waypoint_index = 10, 12
    assert o.find_closest_waypoint() == 10          # true closest (3) is below the window
    o.x_current[:] = (9.0, 0.0, 0.0)
    assert o.find_closest_waypoint(0, 50) == 50     # clipped above
    assert o.find_closest_waypoint(60, 50) == -1    # empty range


def test_next_waypoint_cadence():
    s = np.zeros((300, 3)); s[:, 0] = np.arange(300) * 0.05
    u = np.zeros((300, 2))
    o = make(path=(s, u), start=(0.0, 0.0, 0.0))
    targets = []
    for k in range(20):
        o.x_current[:] = (0.05 * 3, 0.0, 0.0)       # vehicle parked at waypoint 3
        _, t = o.find_next_waypoint(o.x_current)
        o.target_waypoint_index = t
        targets.append(t)
    # +1 per call for 8 calls, the 9th call re-synchronises to closest + lookahead
    assert targets[:8] == [1, 2, 3, 4, 5, 6, 7, 8]
    # on the 9th call the window starts at last_waypoint_index = 8, so 'closest' is 8, target 9
    assert targets[8] == 9
    assert targets[9:17] == [10, 11, 12, 13, 14, 15, 16, 17]
    assert o.last_waypoint_index == targets[-1]


def test_next_waypoint_far_from_path_widens_window():
    s = np.zeros((400, 3)); s[:, 0] = np.arange(400) * 0.05
    u = np.zeros((400, 2))
    o = make(path=(s, u), start=(0.0, 0.0, 0.0))
    o.last_waypoint_index, o.target_waypoint_index = 100, 100
    o.x_current[:] = (2.0, 0.0, 0.0)                # waypoint 40, 3 m behind the window start
    o._count = 8
    _, t = o.find_next_waypoint(o.x_current)
    # closest in the window is 100 (3 m away > 1.2) -> min_index = 100 - 3*20*1.2 = 28 -> closest 40
    assert o.closest_waypoint_index == 40 and t == 41


def test_reference_packing_quirks():
    s, u = path150()
    o = make()
    yref, yref_e = o.pack_references(10)
    assert np.array_equal(yref_e, s[10])            # terminal reference = FIRST waypoint of the window
    assert np.array_equal(yref[:, :3], s[10:60])
    assert np.array_equal(yref[:, 3], u[10:60, 0])
    assert np.all(yref[:, 4] == 0.0)                # steering reference forced to 0
    yref, yref_e = o.pack_references(len(s) - 5)    # past the end: hold the last waypoint, zero inputs
    assert np.array_equal(yref[:5, :3], s[-5:])
    assert np.array_equal(yref[5:, :3], np.tile(s[-1], (45, 1)))
    assert np.all(yref[5:, 3:] == 0.0)


def test_update_current_states_unwraps_yaw():
    o = make()
    o.closest_waypoint_index = 0
    o.state_refs = o.state_refs.copy(); o.state_refs[0, 2] = 3.0
    o.update_current_states(1.0, 2.0, -3.0)
    assert o.x_current[2] == pytest.approx(-3.0 + 2 * math.pi)
    assert tuple(o.x_current[:2]) == (1.0, 2.0)


def test_nearest_direction_and_transform_point():
    nd = Optimizer.NearestDirection
    assert nd(0.1) == 0.0
    assert nd(1.5) == pytest.approx(math.pi / 2)
    # the unwrap at the end is relative to the NORMALISED yaw (include/optimizer.hpp:241-246, 260-265):
    assert nd(-0.1) == pytest.approx(2 * math.pi)   # -0.1 -> 6.18 -> nearest 2pi, stays 2pi
    assert nd(7.0) == 0.0                           # 7.0 - 2pi = 0.717 -> 0
    assert nd(4.0) == pytest.approx(3 * math.pi / 2)
    p = Optimizer.transform_point(np.array([2.0, 1.0, 0.5]), np.array([1.0, 1.0, math.pi / 2]), np.zeros(3))
    assert p == pytest.approx([0.0, -1.0, math.fmod(0.5 - math.pi / 2, 2 * math.pi)])


def test_closed_loop_tracks_reference_path():
    s, _ = path150()
    o = make()
    errs = []
    for k in range(150):
        assert o.update_and_solve(o.x_current) == 0
        o.integrate_next_states()
        d = np.hypot(s[:, 0] - o.x_current[0], s[:, 1] - o.x_current[1])
        errs.append(d.min())
    assert max(errs) < 0.05                          # stays within 5 cm of the recorded path
    assert o.target_waypoint_index > 100
    assert abs(o.u_current[0]) <= 1.0 and abs(o.u_current[1]) <= 0.4
    assert o.iter == 150 and np.all(o.simU[:150, 0] != 0)


def test_solver_error_keeps_previous_command():
    class Failing(OracleSolver):
        fail = False

        def solve(self, *a, **k):
            r = super().solve(*a, **k)
            if self.fail:
                r["status"][:] = 4
            return r
    s, u = path150()
    o = Optimizer(0.1, 40, 0.5, *s[0], state_refs=s, input_refs=u, solver_factory=Failing)
    assert o.update_and_solve(o.x_current) == 0
    u_prev = o.u_current.copy()
    o._solver.fail = True
    assert o.update_and_solve(o.x_current) == 1
    assert np.array_equal(o.u_current, u_prev) and o.status == 4


def test_park_maneuver_reaches_target():
    s, u = path150()
    o = make(start=(3.0, 2.0, 0.05))
    xs = np.array([0.63, 0.32, 0.0, 0.0, 0.0])      # parking offset in the slot frame (mpc_acados_park.py)
    o.set_up_park(xs)
    assert o.frame1 == pytest.approx([3.0, 2.0, 0.0])
    rc, n = 0, 0
    while n < 400:
        rc = o.update_and_solve_park(xs)
        if rc != 0:
            break
        x = o._solver_park.sim_step(o.x_current.reshape(1, 3), o.u_current.reshape(1, 2))[0]
        o.x_current = x
        n += 1
    assert rc == 2, (rc, n, o.x_current_transformed)
    assert np.sum((o.x_current_transformed - xs[:3]) ** 2) < 0.0025


def test_run_reaches_the_end_of_a_short_path():
    """Optimizer::run (src/optimizer.cpp:177-206): loop until within 0.1 of the final waypoint."""
    s, u = path150()
    s, u = s[:90].copy(), u[:91].copy()
    o = make(path=(s, u))
    n = o.run(max_ticks=400)
    assert n < 400
    assert np.linalg.norm(s[-1] - o.x_current) < 0.1
    assert abs(o.t0 - n * o.T) < 1e-9


def test_shift_warm_start_option_shifts_the_solver_iterate():
    """shift_warm_start=True: after a tick the stand-in's stored iterate is the
    solution moved one stage down; off (the reference) it is the solution itself."""
    s, u = path150()
    a = Optimizer(0.1, 40, 0.5, *s[0], state_refs=s, input_refs=u, solver_factory=OracleSolver)
    b = Optimizer(0.1, 40, 0.5, *s[0], state_refs=s, input_refs=u, solver_factory=OracleSolver, shift_warm_start=True)
    assert a.update_and_solve(a.x_current) == 0 and b.update_and_solve(b.x_current) == 0
    assert np.array_equal(a.u_current, b.u_current)
    xa, xb = a._solver.xtraj[0], b._solver.xtraj[0]
    assert np.array_equal(xb[:-1], xa[1:]) and np.array_equal(xb[-1], xa[-1])
    ua, ub = a._solver.utraj[0], b._solver.utraj[0]
    assert np.array_equal(ub[:-1], ua[1:]) and np.array_equal(ub[-1], ua[-1])
