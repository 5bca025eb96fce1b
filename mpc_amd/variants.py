"""OCP variants baked by the reference's code generator, as a ctypes mirror of
``mpc_ocp_spec_t`` (include/mpc_batch.h).

Values follow SURVEY.md §2.1, i.e. the generated set-up code of each variant:
  mobile_robot     src/c_generated_code/acados_solver_mobile_robot.c:380,433-437,474-476,538-541,568-571
  mobile_robot_18  src/c_generated_code_18/acados_solver_mobile_robot_18.c (same lines)
  mobile_robot_25  src/c_generated_code_25/acados_solver_mobile_robot_25.c:354,398-402,501-504,531-534
  mobile_robot_50  src/c_generated_code_50/acados_solver_mobile_robot_50.c (N=80 per its .h:63)
  park             src/c_generated_code_park/acados_solver_park.c:354,389-414,501-504,531-534
The suffixes _18/_25/_50 are reference speeds in cm/s, not horizons.
"""
import ctypes

HPIPM_SPEED_ABS, HPIPM_SPEED, HPIPM_BALANCE, HPIPM_ROBUST = 0, 1, 2, 3

SUCCESS, NAN_DETECTED, MAXITER, MINSTEP, QP_FAILURE = 0, 1, 2, 3, 4

MAX_N = 127


class OcpSpec(ctypes.Structure):
    _fields_ = [
        ("N", ctypes.c_int),
        ("Ts", ctypes.c_double),
        ("L", ctypes.c_double),
        ("W", ctypes.c_double * 5),
        ("W_e", ctypes.c_double * 3),
        ("lbu", ctypes.c_double * 2),
        ("ubu", ctypes.c_double * 2),
        ("lbx", ctypes.c_double * 2),
        ("ubx", ctypes.c_double * 2),
        ("sqp_max_iter", ctypes.c_int),
        ("qp_iter_max", ctypes.c_int),
        ("hpipm_mode", ctypes.c_int),
        ("nlp_tol", ctypes.c_double),
    ]

    def copy(self, **overrides):
        s = OcpSpec.from_buffer_copy(self)
        for k, v in overrides.items():
            cur = getattr(s, k)
            if hasattr(cur, "__len__"):
                for i, x in enumerate(v):
                    cur[i] = x
            else:
                setattr(s, k, v)
        return s


# name -> (N, Ts, W, W_e, lbu, ubu, lbx, ubx, cruise speed used by the synthetic workload)
_TABLE = {
    "mobile_robot": (50, 0.05, (2, 2, 0.5, 1, 0.1), (2, 2, 0.5), (-0.25, -0.42), (0.375, 0.42), (-2, -2), (22, 16), 0.35),
    "mobile_robot_18": (40, 0.125, (2, 2, 0.5, 1, 0.1), (2, 2, 0.5), (-0.4, -0.42), (0.4, 0.42), (-1, -1), (22, 16), 0.18),
    "mobile_robot_25": (40, 0.1, (2, 2, 0.5, 1, 0.25), (2, 2, 0.5), (-1, -0.4), (1, 0.4), (-1, -1), (22, 16), 0.25),
    "mobile_robot_50": (80, 0.05, (2, 2, 0.5, 1, 0.25), (2, 2, 0.5), (-1.5, -0.4), (1.5, 0.4), (-1, -1), (22, 16), 0.5),
    "park": (49, 0.05, (1, 7, 1, 0.5, 0.05), (1, 7, 1), (-0.5, -0.4), (0.5, 0.4), (-4, -4), (4, 4), 0.0),
}

VARIANTS = tuple(_TABLE)


def spec(name="mobile_robot", **overrides):
    """Return the OcpSpec of a generated variant; keyword overrides replace fields
    (e.g. ``spec("mobile_robot_25", N=25)`` for the benchmark horizon of config 2)."""
    N, Ts, W, We, lbu, ubu, lbx, ubx, _ = _TABLE[name]
    s = OcpSpec()
    s.N, s.Ts, s.L = N, Ts, 0.0  # L = 0 selects the generated literals
    s.W[:] = W
    s.W_e[:] = We
    s.lbu[:] = lbu
    s.ubu[:] = ubu
    s.lbx[:] = lbx
    s.ubx[:] = ubx
    s.sqp_max_iter = 1          # SQP_RTI (acados_solver_mobile_robot.c:148)
    s.qp_iter_max = 50          # :682
    s.hpipm_mode = HPIPM_BALANCE  # :670
    s.nlp_tol = 1e-6            # scripts/mobile_robot_18_acados_ocp.json:496-499
    return s.copy(**overrides) if overrides else s


def cruise_speed(name):
    return _TABLE[name][8]
