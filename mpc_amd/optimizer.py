"""Host-side mirror of the reference's MPC front-end `Optimizer`
(include/optimizer.hpp:25-457, src/optimizer.cpp) on top of the batched GPU solver.

Same names, argument meaning and return codes as the C++ class (the C++ path is the contract;
the Python twin scripts/mpc_acados.py differs in two heuristics).  What changes
is only what sits behind `update_and_solve`: instead of N+3 string-keyed acados setter calls and
`mobile_robot_acados_solve` (src/optimizer.cpp:251-296) one `mpc_batch_solve` call is made on a
handle of capacity 1.  Quirks of the reference that are reproduced on purpose:
  * terminal reference = the FIRST waypoint of the window (src/optimizer.cpp:244-251);
  * steering reference forced to 0 (:271); past the end of the path the last waypoint is held
    with zero input references (:255-261);
  * the closest-waypoint scan runs DOWNWARDS with a strict '<', so ties go to the larger
    index (:362-372); window [last_waypoint_index, target + floor(3.95/(v_ref*T))] (:352-356);
  * the target advances by one per tick and re-synchronises to closest+1 on every 9th call
    (:419-425); more than 1.2 m from the path widens the window (:406-410);
  * on a solver error the previous command stays in force (:290-293).
The reference keeps `limit`, `lookahead`, `last_state`, `count` in function-statics
(:387-403), which makes it non-reentrant; here they are per-instance members.
"""
import math
import time

import numpy as np

from . import variants


def load_txt(filename):
    """Optimizer::loadTxt (src/optimizer.cpp:545-577): whitespace-separated rows, all rows the
    same length, row-major."""
    rows = []
    ncols = -1
    with open(filename) as f:
        for line in f:
            vals = [float(v) for v in line.split()]
            if ncols == -1:
                ncols = len(vals)
            elif len(vals) != ncols:
                raise RuntimeError("Inconsistent number of columns")
            rows.append(vals)
    return np.array(rows, dtype=np.float64).reshape(len(rows), max(ncols, 0))


def variant_for_speed(v_ref_int):
    """Solver variant `Optimizer` selects from the reference speed in cm/s (src/optimizer.cpp:34-114)."""
    if v_ref_int >= 30:
        return "mobile_robot"
    if v_ref_int == 25:
        return "mobile_robot_25"
    if v_ref_int == 18:
        return "mobile_robot_18"
    raise ValueError("Invalid reference speed, please use 18, 25 or 50")


def _gpu_solver(spec):
    from . import capi  # raises if libmpc_amd.so / a GPU is missing: no CPU fallback
    return capi.BatchSolver(spec, 1)


class Optimizer:
    """Drop-in counterpart of the C++ `Optimizer`.  `solver_factory(spec)` must return an object
    with solve(x0, yref, yref_e) -> dict(u0, status), sim_step(x, u), reset(); it defaults to
    the GPU library.  Path arrays are given directly (`state_refs` [M,3], `input_refs` [M,2]) or
    loaded from `path_prefix` + {state_refs,input_refs,wp_attributes,wp_normals}<suffix>.txt the
    way the constructor does (src/optimizer.cpp:138-160)."""

    rdb_circumference = 3.95

    def __init__(self, T, N, v_ref, x_init=0.83, y_init=0.33, yaw_init=1.65756, *, state_refs=None,
                 input_refs=None, state_attributes=None, normals=None, paths_dir=None, path_name="_path1",
                 solver_factory=None, shift_warm_start=False):
        # shift_warm_start: shift the solver's iterate by one stage after every tick (SURVEY §8f-2;
        # off = the reference, which keeps the previous iterate as it is)
        self.shift_warm_start = bool(shift_warm_start)
        self.T, self.v_ref = float(T), float(v_ref)
        self.density = 1.0 / self.T / self.v_ref
        self.region_of_acceptance = 0.03076923 * 3 * (0.125 * 1.3) / self.density
        self.t0 = 0.0
        self.v_ref_int = int(self.v_ref * 100)
        self.status = 0
        self.debug = True
        factory = solver_factory or _gpu_solver
        self._factory = factory
        self.variant = variant_for_speed(self.v_ref_int)
        self.use25 = self.variant == "mobile_robot_25"
        self.use18 = self.variant == "mobile_robot_18"
        self.spec = variants.spec(self.variant)
        self.spec_park = variants.spec("park")
        self._solver = factory(self.spec)
        self._solver_park = factory(self.spec_park)
        # dims come from the solver, not from the constructor arguments (src/optimizer.cpp:117-123)
        self.N, self.nx, self.nu = int(self.spec.N), 3, 2
        self.N_park, self.nx_park, self.nu_park = int(self.spec_park.N), 3, 2
        self.x_current = np.array([x_init, y_init, yaw_init], dtype=np.float64)
        self.x_real = self.x_current.copy()
        self.u_current = np.zeros(2)
        self.x_state = np.zeros(5)
        if state_refs is None:
            sfx = path_name + str(self.v_ref_int) + ".txt"
            state_refs = load_txt(f"{paths_dir}/state_refs{sfx}")
            input_refs = load_txt(f"{paths_dir}/input_refs{sfx}")
            state_attributes = load_txt(f"{paths_dir}/wp_attributes{sfx}").ravel()
            normals = load_txt(f"{paths_dir}/wp_normals{sfx}")
        self.state_refs = np.ascontiguousarray(state_refs, np.float64)
        self.input_refs = np.ascontiguousarray(input_refs, np.float64)
        self.state_attributes = None if state_attributes is None else np.asarray(state_attributes, np.float64).ravel()
        self.normals = normals
        self.left_turn_states = self.right_turn_states = self.straight_states = None
        self.state_refs_ptr = self.state_refs
        self.num_waypoints = self.state_refs.shape[0]
        self.target_waypoint_index = 0
        self.last_waypoint_index = 0
        self.closest_waypoint_index = 0
        self.iter = 0
        # per-tick record behind computeStats (the reference's simX / simU / *_errors buffers, which
        # it sizes 2*num_waypoints and overruns on longer runs; these grow)
        self._simX, self._simU, self._err, self._time_ms = [], [], [], []
        self.results_dir = None
        # computeStats: True = the reference's zero-padded means and its time_record arithmetic
        self.reference_stats = True
        self.time_stats = np.zeros(3)          # [mean, max, min] tick time in ms, set by computeStats
        # integrate_next_states for the 18 cm/s variant: see there
        self.use18_plant_as_written = False
        self._solver_plant50 = None
        if self.v_ref > 0.3:
            self.v_ref = 0.35  # src/optimizer.cpp:174
        # per-instance versions of the function-statics of find_next_waypoint / find_closest_waypoint
        self._limit = None
        self._lookahead = 1
        self._count = 0
        self._last_state = None
        self._park_count = 0
        self.frame1 = np.zeros(3)
        self.frame2 = np.zeros(3)
        self.x_current_transformed = np.zeros(3)

    # ------------------------------------------------------------------ waypoint logic (H2)
    def _limit_value(self):
        if self._limit is None:
            self._limit = int(math.floor(self.rdb_circumference / (self.v_ref * self.T)))
        return self._limit

    def find_closest_waypoint(self, min_index=-1, max_index=-1):
        """src/optimizer.cpp:343-382."""
        rows = self.state_refs.shape[0]
        limit = self._limit_value()
        if min_index < 0:
            min_index = min(self.last_waypoint_index, rows - 1)
        if max_index < 0:
            max_index = min(self.target_waypoint_index + limit, rows - 1)
        current_norm = self.x_current[0] * self.x_current[0] + self.x_current[1] * self.x_current[1]
        min_distance_sq = np.finfo(np.float64).max
        closest = -1
        for i in range(max_index, min_index - 1, -1):
            px, py = self.state_refs[i, 0], self.state_refs[i, 1]
            distance_sq = (px * px + py * py) - 2 * (px * self.x_current[0] + py * self.x_current[1]) + current_norm
            if distance_sq < min_distance_sq:
                min_distance_sq = distance_sq
                closest = i
        return closest

    def find_next_waypoint(self, i_current_state, safety_check=True, min_index=-1, max_index=-1):
        """src/optimizer.cpp:384-431.  Returns (1, output_target) and updates the indices."""
        self._limit_value()
        if self.v_ref > 0.375:
            self._lookahead = 1
        self._last_state = np.array(i_current_state, dtype=np.float64)
        self.closest_waypoint_index = self.find_closest_waypoint(min_index, max_index)
        c = self.closest_waypoint_index
        dx = self.state_refs[c, 0] - i_current_state[0]
        dy = self.state_refs[c, 1] - i_current_state[1]
        distance_to_current = math.sqrt(dx * dx + dy * dy)
        if distance_to_current > 1.2:
            min_index = int(max(c - distance_to_current * self.density * 1.2, 0.0))
            self.closest_waypoint_index = self.find_closest_waypoint(min_index, max_index)
        if self._count >= 8:
            target = self.closest_waypoint_index + self._lookahead
            self._count = 0
        else:
            target = self.target_waypoint_index + 1
            self._count += 1
        output_target = min(target, self.state_refs_ptr.shape[0] - 1)
        self.last_waypoint_index = output_target
        return 1, output_target

    def pack_references(self, target):
        """yref [N,5] and yref_e [3] of one tick (src/optimizer.cpp:240-275)."""
        refs = self.state_refs_ptr
        rows = refs.shape[0]
        idx = target if target < self.state_refs.shape[0] else self.state_refs.shape[0] - 1
        yref_e = refs[target, :3].copy()
        yref = np.zeros((self.N, 5))
        for j in range(self.N):
            if target + j >= rows:
                yref[j, :3] = refs[rows - 1, :3]
            else:
                yref[j, :3] = refs[idx + j, :3]
                yref[j, 3:5] = self.input_refs[idx + j, :2]
            yref[j, 4] = 0.0  # "set steer rate to 0"
        return yref, yref_e

    # ------------------------------------------------------------------ the tick (H1)
    def update_and_solve(self, i_current_state=None, safety_check=True, mode=-1):
        """Returns 0 if successful, 1 otherwise; the control is left in `u_current`."""
        t_start = time.perf_counter()   # debug timer (src/optimizer.cpp:214)
        if i_current_state is None:
            i_current_state = self.x_current
        if mode == -1:
            self.state_refs_ptr = self.state_refs
        elif mode == 0 and self.left_turn_states is not None:
            self.state_refs_ptr = self.left_turn_states
        elif mode == 1 and self.straight_states is not None:
            self.state_refs_ptr = self.straight_states
        elif mode == 2 and self.right_turn_states is not None:
            self.state_refs_ptr = self.right_turn_states
        else:
            self.state_refs_ptr = self.state_refs
        _, self.target_waypoint_index = self.find_next_waypoint(i_current_state, safety_check)
        yref, yref_e = self.pack_references(self.target_waypoint_index)
        self.x_state[:3] = yref[self.N - 1, :3]
        x0 = np.asarray(i_current_state, np.float64).reshape(1, 3)
        out = self._solver.solve(x0, yref[None], yref_e[None])
        if self.shift_warm_start:
            self._solver.shift()
        self.status = int(out["status"][0])
        if self.status != 0:
            return 1
        self.u_current = out["u0"][0].copy()
        if self.debug:
            self._record_tick(i_current_state, 1e3 * (time.perf_counter() - t_start))
        return 0

    def _record_tick(self, state, elapsed_ms=0.0):
        """src/optimizer.cpp:300-317 / include/optimizer.hpp:223-236."""
        self._time_ms.append(float(elapsed_ms))
        t = min(self.target_waypoint_index, self.state_refs.shape[0] - 1)
        self._simU.append(np.array(self.u_current, np.float64))
        self._simX.append(np.array(state, np.float64))
        self._err.append(self.state_refs[t, :3] - self.x_real if t >= 0 else np.zeros(3))
        self.iter += 1

    @property
    def simX(self):
        return np.array(self._simX).reshape(-1, 3)

    @property
    def simU(self):
        return np.array(self._simU).reshape(-1, 2)

    @property
    def time_record(self):
        """[iter, nu] like the reference's (src/optimizer.cpp:172,304,475): ms in column 0, column 1 never written."""
        t = np.zeros((len(self._time_ms), 2))
        t[:, 0] = self._time_ms
        return t

    def computeStats(self, hsy=0):
        """src/optimizer.cpp:472-518: [mean speed, mean steer, mean |d speed|, mean |d steer|,
        mean |x err|, mean |y err|, mean |yaw err|]; the tick timing of :476-478 goes to `time_stats`
        (mean, max, min in ms) and is printed.  reference_stats (default): the reference's numbers —
        its error buffers are 2*num_waypoints long, zero beyond the recorded ticks, and .mean() runs
        over all of them (:166-171, 496-499); its time_record is resized to [iter, nu] with only the
        first column ever written (:172, 304, 475), so its "average" is sum / (nu*iter) and its "min" 0.
        reference_stats = False: means over the recorded ticks only.  Files only when `results_dir`
        is set."""
        n = self.iter
        stats = np.zeros(7)
        self.time_stats = np.zeros(3)
        if n == 0:
            return stats
        U, E = self.simU, np.array(self._err).reshape(-1, 3)
        stats[0], stats[1] = U[:, 0].mean(), U[:, 1].mean()
        if n > 1:
            dU = np.abs(np.diff(U, axis=0))
            stats[2], stats[3] = dU[:, 0].mean(), dU[:, 1].mean()
        nerr = 2.0 * self.num_waypoints if (self.reference_stats and self.num_waypoints > 0) else float(n)
        stats[4], stats[5] = np.abs(E[:, 0]).sum() / nerr, np.abs(E[:, 1]).sum() / nerr
        stats[6] = np.abs(np.arctan2(np.sin(E[:, 2]), np.cos(E[:, 2]))).sum() / nerr
        tm = np.asarray(self._time_ms)
        if self.reference_stats:
            self.time_stats = np.array([tm.sum() / (2.0 * n), tm.max(), 0.0])
        else:
            self.time_stats = np.array([tm.mean(), tm.max(), tm.min()])
        print("average estimation time %f ms " % self.time_stats[0])
        print("max estimation time %f ms " % self.time_stats[1])
        print("min estimation time %f ms " % self.time_stats[2])
        if self.results_dir:
            np.savetxt(f"{self.results_dir}/simX.txt", self.simX)
            np.savetxt(f"{self.results_dir}/simU.txt", U)
            np.savetxt(f"{self.results_dir}/stats.txt", stats)
            np.savetxt(f"{self.results_dir}/state_refs.txt", self.state_refs)
        return stats

    # ------------------------------------------------------------------ waypoint attributes
    (NORMAL, CROSSWALK, INTERSECTION, ONEWAY, HIGHWAYLEFT, HIGHWAYRIGHT, ROUNDABOUT, STOPLINE, DOTTED,
     DOTTED_CROSSWALK) = range(10)   # enum ATTRIBUTE, include/optimizer.hpp:94-96

    def attribute_cmp(self, idx, attr):
        """include/optimizer.hpp:97-102: attribute, or the same + 100 (not detectable there)."""
        if idx < 0 or idx >= self.state_attributes.shape[0]:
            return False
        a = self.state_attributes[idx]
        return bool(a == attr or a == attr + 100)

    def is_not_detectable(self, idx):
        """include/optimizer.hpp:103-105."""
        return bool(self.state_attributes[idx] >= 100 or self.attribute_cmp(idx, self.DOTTED_CROSSWALK)
                    or self.attribute_cmp(idx, self.INTERSECTION) or self.attribute_cmp(idx, self.ROUNDABOUT))

    def get_current_attribute(self):
        return int(self.state_attributes[self.target_waypoint_index])

    def get_current_waypoints(self):
        """include/optimizer.hpp:443-445."""
        t = self.target_waypoint_index
        if t + self.N > self.state_refs.shape[0]:
            raise IndexError("get_current_waypoints: window past the end of state_refs")
        return self.state_refs[t:t + self.N, :3].copy()

    def change_lane(self, start_index, end_index, shift_right=False, shift_distance=0.36 - 0.1):
        """include/optimizer.hpp:335-372: shift [start, end) along the normals, ramped over
        int(density / 8) waypoints at both ends."""
        if shift_right:
            shift_distance *= -1
        ramp = int(self.density * 0.125)
        up_end, down_start = start_index + ramp, end_index - ramp
        for i in range(start_index, end_index):
            shift = shift_distance
            if i < up_end:
                shift = shift_distance * ((i - start_index + 1) / ramp)
            elif i >= down_start:
                shift = shift_distance * ((end_index - i) / ramp)
            self.state_refs[i, :2] += np.asarray(self.normals)[i, :2] * shift

    @staticmethod
    def NearestDirectionIndex(yaw):
        """include/optimizer.hpp:268-294: 0 east, 1 north, 2 west, 3 south."""
        while yaw > 2 * math.pi:
            yaw -= 2 * math.pi
        while yaw < 0:
            yaw += 2 * math.pi
        best, best_d = 0, abs(yaw)
        for i in range(1, 5):
            d = abs(yaw - i * (math.pi / 2))
            if d < best_d:
                best, best_d = i, d
        return 0 if best == 4 else best

    def integrate_next_states(self):
        """src/optimizer.cpp:321-341 (H16): one ERK4 step of the variant's Tsim.  For the 18 cm/s
        variant the reference's else-branch (:328-332) names the N = 50 variant's integrator
        (Ts = 0.05) on a capsule its use18 constructor branch never created — undefined behaviour
        that cannot be reproduced; `use18_plant_as_written` takes the branch by its letter, as if
        that capsule had been created (the plant then advances 0.05 s per tick)."""
        solver = self._solver
        if self.use18 and self.use18_plant_as_written:
            if self._solver_plant50 is None:
                self._solver_plant50 = self._factory(variants.spec("mobile_robot"))
            solver = self._solver_plant50
        self.x_current = solver.sim_step(self.x_current.reshape(1, 3), self.u_current.reshape(1, 2))[0]
        self.t0 += self.T

    def update_current_states(self, x, y, yaw, state=None, safety_check=True):
        """src/optimizer.cpp:432-455: unwrap yaw towards the closest waypoint's heading."""
        if state is None:
            state = self.x_current
        rows = self.state_refs.shape[0]
        if self.closest_waypoint_index < rows and rows > 0:
            ref_yaw = self.state_refs[self.closest_waypoint_index, 2]
            while ref_yaw - yaw > math.pi:
                yaw += 2 * math.pi
            while ref_yaw - yaw < -math.pi:
                yaw -= 2 * math.pi
        state[0], state[1], state[2] = x, y, yaw
        return 1

    def initialize_current_states(self, x, y, yaw):
        return self.update_current_states(x, y, yaw, self.x_current, False)

    def reset_solver(self):
        """include/optimizer.hpp:446-456."""
        self._solver.reset()
        return 0

    def run(self, max_ticks=None):
        """Offline closed loop, src/optimizer.cpp:177-206."""
        x_final = self.state_refs[-1]
        hsy = 0
        while True:
            error_norm = float(np.linalg.norm(x_final - self.x_current))
            if self.target_waypoint_index > self.num_waypoints or error_norm < 0.1:
                break
            if max_ticks is not None and hsy >= max_ticks:
                break
            self.update_and_solve(self.x_current)
            self.integrate_next_states()
            hsy += 1
        return hsy

    # ------------------------------------------------------------------ parking (H17)
    @staticmethod
    def NearestDirection(yaw):
        """include/optimizer.hpp:240-267."""
        while yaw > 2 * math.pi:
            yaw -= 2 * math.pi
        while yaw < 0:
            yaw += 2 * math.pi
        directions = [0, math.pi / 2, math.pi, 3 * math.pi / 2, 2 * math.pi]
        min_difference = abs(yaw - directions[0])
        nearest = directions[0]
        for d in directions[1:]:
            if abs(yaw - d) < min_difference:
                min_difference = abs(yaw - d)
                nearest = d
        while nearest - yaw > math.pi:
            nearest -= 2 * math.pi
        while nearest - yaw < -math.pi:
            nearest += 2 * math.pi
        return nearest

    @staticmethod
    def transform_point(pt, frame1, frame2):
        """include/optimizer.hpp:140-166."""
        x1, y1, theta1 = frame1
        x2, y2, theta2 = frame2
        xt, yt = pt[0] - x1, pt[1] - y1
        rot = theta2 - theta1
        c, s = math.cos(rot), math.sin(rot)
        rx, ry = c * xt - s * yt, s * xt + c * yt
        psi_t = math.fmod(pt[2] + rot, 2 * math.pi)
        return np.array([rx + x2, ry + y2, psi_t])

    def set_up_park(self, xs):
        """include/optimizer.hpp:167-186: reset the park capsule, constant target on all stages."""
        self._solver_park.reset()
        self._park_xs = np.asarray(xs, np.float64).copy()
        self.frame1 = np.array([self.x_current[0], self.x_current[1], self.NearestDirection(self.x_current[2])])
        self.frame2 = np.zeros(3)

    def update_and_solve_park(self, xs, thresh_sq=0.0025):
        """include/optimizer.hpp:191-238.  0 ok, 1 solver error, 2 target reached."""
        t_start = time.perf_counter()   # include/optimizer.hpp:192
        xs = np.asarray(xs, np.float64)
        xt = self.transform_point(self.x_current, self.frame1, self.frame2)
        yaw_frame2 = self.frame2[2]
        while xt[2] - yaw_frame2 > math.pi:
            xt[2] -= 2 * math.pi
        while xt[2] - yaw_frame2 < -math.pi:
            xt[2] += 2 * math.pi
        self.x_current_transformed = xt
        error_sq = float(np.sum((xt - xs[:3]) ** 2))
        self._park_count += 1
        if error_sq < thresh_sq:
            self._park_count = 0
            return 2
        yref = np.tile(self._park_xs[:5], (self.N_park, 1))
        out = self._solver_park.solve(xt.reshape(1, 3), yref[None], self._park_xs[None, :3])
        self.status = int(out["status"][0])
        if self.status != 0:
            return 1
        self.u_current = out["u0"][0].copy()
        if self.debug:
            self._record_tick(self.x_current, 1e3 * (time.perf_counter() - t_start))
        return 0

    def close(self):
        for s in (self._solver, self._solver_park, self._solver_plant50):
            if hasattr(s, "close"):
                s.close()
