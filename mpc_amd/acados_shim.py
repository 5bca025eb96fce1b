"""`AcadosOcpSolver` / `AcadosSimSolver` as the reference's Python scripts use them
(scripts/mpc_acados.py:236-238 create, :250-264 set/solve/get, :270-276 integrator; the same calls
in mpc_acados2.py, mpc_acados3.py, mpc_acados_park.py), bound with ctypes to the acados-named entry
points libmpc_amd.so exports (include/acados_compat/, mpc_amd/csrc/acados_compat.cpp) — the Python
half of "relink, don't rewrite": a script that does

    from mpc_amd.acados_shim import AcadosOcpSolver, AcadosSimSolver
    solver = AcadosOcpSolver(ocp, json_file="mobile_robot_acados_ocp.json")
    integrator = AcadosSimSolver(ocp, json_file="mobile_robot_acados_ocp.json")

keeps its `solver.set(j, 'yref', ...)`, `solver.set(0, 'lbx', ...)`, `solver.solve()`,
`solver.get(0, 'u')`, `solver.reset()`, `integrator.set('x', ...)`, `integrator.solve()`,
`integrator.get('x')` lines as they are.  No code is generated and nothing is compiled: the solver
is one of the five variants the library carries (mpc_spec_variant), chosen by the model name —
`ocp.model.name` if the object has one, else the `<name>_acados_ocp.json` file name, else `name=`.
An OCP description that does not match the variant's constants cannot be honoured and is not
inspected; setters for anything but the fields below fail loudly, as acados_template's do for
unknown fields.  There is no CPU fallback: without the HIP library (or without a GPU) construction
raises.
"""
import ctypes
import os

import numpy as np

from . import capi

VARIANTS = ("mobile_robot", "mobile_robot_18", "mobile_robot_25", "mobile_robot_50", "park")
_vp = ctypes.c_void_p


def resolve_name(acados_ocp=None, json_file=None, name=None):
    """Which generated solver the caller means (module docstring)."""
    if name is None and acados_ocp is not None:
        model = getattr(acados_ocp, "model", None)
        name = getattr(model, "name", None)
    if name is None and json_file is not None:
        base = os.path.basename(str(json_file))
        if base.endswith("_acados_ocp.json"):
            name = base[: -len("_acados_ocp.json")]
        elif base.endswith(".json"):
            name = base[: -len(".json")]
    if name not in VARIANTS:
        raise ValueError(f"acados_shim: no solver variant named {name!r} in libmpc_amd.so (have: {', '.join(VARIANTS)})")
    return name


def _fn(name, restype, *argtypes):
    f = getattr(capi.lib(), name)
    f.restype = restype
    f.argtypes = list(argtypes)
    return f


def _darr(value, n, what):
    a = np.ascontiguousarray(np.asarray(value, dtype=np.float64).reshape(-1))
    if a.size != n:
        raise Exception(f"acados_shim: {what} takes {n} values, got {a.size}")
    return a


class AcadosOcpSolver:
    """acados_template.AcadosOcpSolver over `<name>_acados_*` (acados_solver_<name>.h:122-161)."""

    def __init__(self, acados_ocp=None, json_file=None, name=None, **_ignored):
        self.name = n = resolve_name(acados_ocp, json_file, name)
        self._create_capsule = _fn(f"{n}_acados_create_capsule", _vp)
        self._create = _fn(f"{n}_acados_create", ctypes.c_int, _vp)
        self._solve = _fn(f"{n}_acados_solve", ctypes.c_int, _vp)
        self._reset = _fn(f"{n}_acados_reset", ctypes.c_int, _vp, ctypes.c_int)
        self._free = _fn(f"{n}_acados_free", ctypes.c_int, _vp)
        self._free_capsule = _fn(f"{n}_acados_free_capsule", ctypes.c_int, _vp)
        self._print_stats = _fn(f"{n}_acados_print_stats", None, _vp)
        self._cost_set = _fn("ocp_nlp_cost_model_set", ctypes.c_int, _vp, _vp, _vp, ctypes.c_int, ctypes.c_char_p, _vp)
        self._con_set = _fn("ocp_nlp_constraints_model_set", ctypes.c_int, _vp, _vp, _vp, ctypes.c_int, ctypes.c_char_p, _vp)
        self._out_set = _fn("ocp_nlp_out_set", None, _vp, _vp, _vp, ctypes.c_int, ctypes.c_char_p, _vp)
        self._out_get = _fn("ocp_nlp_out_get", None, _vp, _vp, _vp, ctypes.c_int, ctypes.c_char_p, _vp)
        self._dims_attr = _fn("ocp_nlp_dims_get_from_attr", ctypes.c_int, _vp, _vp, _vp, ctypes.c_int, ctypes.c_char_p)
        self._nlp_get = _fn("ocp_nlp_get", None, _vp, _vp, ctypes.c_char_p, _vp)
        self._opts_set = _fn("ocp_nlp_solver_opts_set", None, _vp, _vp, ctypes.c_char_p, _vp)
        self.capsule = self._create_capsule()
        if not self.capsule:
            raise capi.MpcError(f"{n}_acados_create_capsule failed")
        rc = self._create(self.capsule)
        if rc != 0:
            self._free_capsule(self.capsule)
            self.capsule = None
            raise capi.MpcError(f"{n}_acados_create returned {rc} (no GPU, or the HIP library failed to set up: see stderr)")
        g = lambda what: _fn(f"{n}_acados_get_nlp_{what}", _vp, _vp)(self.capsule)  # noqa: E731
        self.nlp_config, self.nlp_dims, self.nlp_in, self.nlp_out = g("config"), g("dims"), g("in"), g("out")
        self.nlp_solver, self.nlp_opts = g("solver"), g("opts")
        spec = capi.spec_variant(n)
        self.N = spec.N
        self.status = 0

    # -- acados_template.AcadosOcpSolver.set(stage, field, value) --
    def set(self, stage, field, value_):
        stage = int(stage)
        f = field.encode()
        if field in ("yref", "y_ref", "W"):
            n = {"yref": 5 if stage < self.N else 3, "y_ref": 5 if stage < self.N else 3, "W": 25 if stage < self.N else 9}[field]
            a = _darr(value_, n, f"set({stage}, '{field}')")
            rc = self._cost_set(self.nlp_config, self.nlp_dims, self.nlp_in, stage, f, a.ctypes.data)
        elif field in ("lbx", "ubx", "lbu", "ubu"):
            n = 3 if (stage == 0 and field in ("lbx", "ubx")) else 2
            a = _darr(value_, n, f"set({stage}, '{field}')")
            rc = self._con_set(self.nlp_config, self.nlp_dims, self.nlp_in, stage, f, a.ctypes.data)
        elif field in ("x", "u", "pi"):
            n = self._dims_attr(self.nlp_config, self.nlp_dims, self.nlp_out, stage, f)
            a = _darr(value_, n, f"set({stage}, '{field}')")
            self._out_set(self.nlp_config, self.nlp_dims, self.nlp_out, stage, f, a.ctypes.data)
            rc = 0
        else:
            raise Exception(f"AcadosOcpSolver.set(): '{field}' is not a field this solver can set "
                            "(have: yref, W, lbx, ubx, lbu, ubu, x, u, pi)")
        if rc != 0:
            raise Exception(f"AcadosOcpSolver.set({stage}, '{field}') was refused (status {rc}): this solver carries the "
                            "generated values of everything but yref and the initial state")

    def cost_set(self, stage, field, value_):
        self.set(stage, field, value_)

    def constraints_set(self, stage, field, value_):
        self.set(stage, field, value_)

    # -- get(stage, field) --
    def get(self, stage, field):
        if field not in ("x", "u", "pi"):
            raise Exception(f"AcadosOcpSolver.get(): '{field}' is not available (have: x, u, pi)")
        f = field.encode()
        n = self._dims_attr(self.nlp_config, self.nlp_dims, self.nlp_out, int(stage), f)
        out = np.zeros(n)
        if n:
            self._out_get(self.nlp_config, self.nlp_dims, self.nlp_out, int(stage), f, out.ctypes.data)
        return out

    def solve(self):
        self.status = int(self._solve(self.capsule))
        return self.status

    def solve_for_x0(self, x0_bar, fail_on_nonzero_status=True, print_stats_on_failure=True):
        self.set(0, "lbx", x0_bar)
        self.set(0, "ubx", x0_bar)
        status = self.solve()
        if status != 0:
            if print_stats_on_failure:
                self.print_statistics()
            if fail_on_nonzero_status:
                raise Exception(f"acados acados_ocp_solver returned status {status}")
        return self.get(0, "u")

    def reset(self, reset_qp_solver_mem=1):
        return int(self._reset(self.capsule, int(reset_qp_solver_mem)))

    def get_status(self):
        return self.status

    def get_stats(self, field_):
        f = field_.encode()
        if field_ in ("time_tot", "time_qp", "time_qp_sol", "time_qp_solver_call", "time_lin", "time_reg", "time_sim",
                      "time_glob", "time_sim_la", "time_sim_ad", "time_qp_xcond"):
            v = ctypes.c_double(0.0)
            self._nlp_get(self.nlp_config, self.nlp_solver, f, ctypes.byref(v))
            return np.array([v.value])
        if field_ in ("sqp_iter", "nlp_iter", "qp_iter", "status", "qp_status", "qp_stat", "stat_m", "stat_n"):
            v = ctypes.c_int(0)
            self._nlp_get(self.nlp_config, self.nlp_solver, f, ctypes.byref(v))
            return np.array([v.value]) if field_ in ("sqp_iter", "nlp_iter") else v.value
        if field_ == "statistics":
            m, n = self.get_stats("stat_m"), self.get_stats("stat_n")
            buf = np.zeros((n + 1) * m)
            self._nlp_get(self.nlp_config, self.nlp_solver, f, buf.ctypes.data)
            return buf.reshape(n + 1, m)   # rows: iter, qp_stat, qp_iter (column-major in C)
        raise Exception(f"AcadosOcpSolver.get_stats(): '{field_}' is not available")

    def print_statistics(self):
        self._print_stats(self.capsule)

    def options_set(self, field_, value_):
        if field_ == "rti_phase":
            v = ctypes.c_int(int(value_))
            self._opts_set(self.nlp_config, self.nlp_opts, b"rti_phase", ctypes.byref(v))
            if int(value_) != 0:
                raise Exception("AcadosOcpSolver.options_set('rti_phase'): only 0 (preparation + feedback in one call)")
        elif field_ == "print_level":
            v = ctypes.c_int(int(value_))
            self._opts_set(self.nlp_config, self.nlp_opts, b"print_level", ctypes.byref(v))
        else:
            raise Exception(f"AcadosOcpSolver.options_set(): '{field_}' cannot be changed in this solver")

    def __del__(self):
        cap = getattr(self, "capsule", None)
        if cap:
            try:
                self._free(cap)
                self._free_capsule(cap)
            except Exception:   # noqa: BLE001  (interpreter shutdown)
                pass
            self.capsule = None


class AcadosSimSolver:
    """acados_template.AcadosSimSolver over `<name>_acados_sim_*` (acados_sim_solver_<name>.h):
    one ERK4 step of length Ts of the variant (scripts/mpc_acados.py:270-276)."""

    def __init__(self, acados_sim=None, json_file=None, name=None, **_ignored):
        self.name = n = resolve_name(acados_sim, json_file, name)
        self._create_capsule = _fn(f"{n}_acados_sim_solver_create_capsule", _vp)
        self._create = _fn(f"{n}_acados_sim_create", ctypes.c_int, _vp)
        self._solve = _fn(f"{n}_acados_sim_solve", ctypes.c_int, _vp)
        self._free = _fn(f"{n}_acados_sim_free", ctypes.c_int, _vp)
        self._free_capsule = _fn(f"{n}_acados_sim_solver_free_capsule", ctypes.c_int, _vp)
        self._in_set = _fn("sim_in_set", ctypes.c_int, _vp, _vp, _vp, ctypes.c_char_p, _vp)
        self._out_get = _fn("sim_out_get", ctypes.c_int, _vp, _vp, _vp, ctypes.c_char_p, _vp)
        self.capsule = self._create_capsule()
        if not self.capsule:
            raise capi.MpcError(f"{n}_acados_sim_solver_create_capsule failed")
        rc = self._create(self.capsule)
        if rc != 0:
            self._free_capsule(self.capsule)
            self.capsule = None
            raise capi.MpcError(f"{n}_acados_sim_create returned {rc} (no GPU, or the HIP library failed to set up: see stderr)")
        g = lambda what: _fn(f"{n}_acados_get_sim_{what}", _vp, _vp)(self.capsule)  # noqa: E731
        self.sim_config, self.sim_dims, self.sim_in, self.sim_out = g("config"), g("dims"), g("in"), g("out")

    def set(self, field_, value_):
        n = {"x": 3, "u": 2, "T": 1}.get(field_)
        if n is None:
            raise Exception(f"AcadosSimSolver.set(): '{field_}' is not a field this integrator can set (have: x, u, T)")
        a = _darr(value_, n, f"set('{field_}')")
        rc = self._in_set(self.sim_config, self.sim_dims, self.sim_in, field_.encode(), a.ctypes.data)
        if rc != 0:
            raise Exception(f"AcadosSimSolver.set('{field_}') was refused (status {rc})")

    def get(self, field_):
        if field_ not in ("x", "xn"):
            raise Exception(f"AcadosSimSolver.get(): '{field_}' is not available (have: x)")
        out = np.zeros(3)
        rc = self._out_get(self.sim_config, self.sim_dims, self.sim_out, field_.encode(), out.ctypes.data)
        if rc != 0:
            raise Exception(f"AcadosSimSolver.get('{field_}') was refused (status {rc})")
        return out

    def solve(self):
        return int(self._solve(self.capsule))

    def simulate(self, x=None, u=None):
        if x is not None:
            self.set("x", x)
        if u is not None:
            self.set("u", u)
        status = self.solve()
        if status != 0:
            raise Exception(f"acados integrator returned status {status}")
        return self.get("x")

    def __del__(self):
        cap = getattr(self, "capsule", None)
        if cap:
            try:
                self._free(cap)
                self._free_capsule(cap)
            except Exception:   # noqa: BLE001
                pass
            self.capsule = None
