"""mpc_amd/acados_shim.py: `AcadosOcpSolver` / `AcadosSimSolver` with the calls the reference's
Python scripts make (scripts/mpc_acados.py:236-238, 250-264, 270-276), over the acados-named entry
points of libmpc_amd.so.

CPU: which variant a constructor call means; every entry point the shim binds is exported; without a
GPU construction raises instead of falling back.
GPU: the reference's tick — set yref on every stage, lbx/ubx at stage 0, solve, get u — against the
same inputs through mpc_batch_solve: bit for bit, 20 warm ticks, two variants; the integrator
against mpc_batch_sim_step; reset reproduces the cold start; what acados_template refuses is
refused."""
import types

import numpy as np
import pytest

from mpc_amd import acados_shim, capi, workload


def test_which_variant_a_constructor_call_means():
    r = acados_shim.resolve_name
    ocp = types.SimpleNamespace(model=types.SimpleNamespace(name="mobile_robot_25"))
    assert r(ocp, "./whatever.json") == "mobile_robot_25"                 # ocp.model.name wins (scripts/mpc_acados.py:135)
    assert r(None, "./mobile_robot_acados_ocp.json") == "mobile_robot"    # AcadosOcpSolver(ocp, json_file=...) (:234-236)
    assert r(object(), "/x/park_acados_ocp.json") == "park"
    assert r(None, None, name="mobile_robot_18") == "mobile_robot_18"
    for bad in ((None, "scandy_acados_ocp.json", None), (None, None, None), (None, None, "mobile_robot_99")):
        with pytest.raises(ValueError):
            r(*bad)


def test_every_entry_point_the_shim_binds_is_exported():
    L = capi.lib()
    for v in acados_shim.VARIANTS:
        for s in ("create_capsule", "create", "solve", "reset", "free", "free_capsule", "print_stats", "get_nlp_config",
                  "get_nlp_dims", "get_nlp_in", "get_nlp_out", "get_nlp_solver", "get_nlp_opts", "sim_solver_create_capsule",
                  "sim_create", "sim_solve", "sim_free", "sim_solver_free_capsule", "get_sim_config", "get_sim_dims",
                  "get_sim_in", "get_sim_out"):
            assert hasattr(L, f"{v}_acados_{s}"), (v, s)
    for f in ("ocp_nlp_cost_model_set", "ocp_nlp_constraints_model_set", "ocp_nlp_out_set", "ocp_nlp_out_get",
              "ocp_nlp_dims_get_from_attr", "ocp_nlp_get", "ocp_nlp_solver_opts_set", "sim_in_set", "sim_out_get"):
        assert hasattr(L, f)


def test_construction_fails_loudly_without_a_gpu():
    torch = pytest.importorskip("torch")
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(capi.MpcError):
        acados_shim.AcadosOcpSolver(None, json_file="mobile_robot_acados_ocp.json")


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["mobile_robot", "park"])
def test_reference_tick_through_the_shim_equals_mpc_batch_solve(name):
    sp, wl = workload.make(name, 1)
    N = sp.N
    direct = capi.BatchSolver(sp, 1)
    solver = acados_shim.AcadosOcpSolver(types.SimpleNamespace(model=types.SimpleNamespace(name=name)),
                                         json_file=f"./{name}_acados_ocp.json")
    assert solver.N == N
    for t in range(20):
        x0, yref, yref_e = wl.tick(t)
        r = direct.solve(x0, yref, yref_e)
        xt, ut, pit = direct.get(1)
        # scripts/mpc_acados.py:250-264
        solver.set(N, "yref", yref_e[0])
        for j in range(N):
            solver.set(j, "yref", yref[0, j])
        solver.set(0, "lbx", x0[0])
        solver.set(0, "ubx", x0[0])
        status = solver.solve()
        assert status == int(r["status"][0]) == 0
        assert np.array_equal(solver.get(0, "u"), r["u0"][0]), t
        for j in (0, 1, N // 2, N - 1):
            assert np.array_equal(solver.get(j, "x"), xt[0, j]) and np.array_equal(solver.get(j, "u"), ut[0, j])
            assert np.array_equal(solver.get(j, "pi"), pit[0, j])
        assert np.array_equal(solver.get(N, "x"), xt[0, N]) and solver.get(N, "u").size == 0
        assert solver.get_stats("qp_iter") == int(r["qp_iter"][0])
        assert solver.get_stats("time_tot")[0] > 0.0 and solver.get_stats("sqp_iter")[0] == 1
    # reset -> the cold start again (scripts/mpc_acados.py:  self.solver.reset())
    x0, yref, yref_e = wl.tick(0)
    cold = capi.BatchSolver(sp, 1).solve(x0, yref, yref_e)
    assert solver.reset() == 0
    solver.set(N, "yref", yref_e[0])
    for j in range(N):
        solver.set(j, "yref", yref[0, j])
    assert np.array_equal(solver.solve_for_x0(x0[0]), cold["u0"][0])
    # what this solver cannot honour is refused, not ignored
    with pytest.raises(Exception):
        solver.set(0, "p", np.zeros(1))
    with pytest.raises(Exception):
        solver.set(3, "yref", np.zeros(3))          # 5 values on a path stage
    with pytest.raises(Exception):
        solver.set(2, "ubu", np.array([9.0, 9.0]))  # not the generated bound
    solver.set(2, "ubu", np.array(sp.ubu))          # the generated bound is accepted
    with pytest.raises(Exception):
        solver.get(0, "sl")
    with pytest.raises(Exception):
        solver.options_set("qp_tol_stat", 1e-3)
    direct.close()


@pytest.mark.gpu
def test_integrator_through_the_shim_equals_sim_step():
    sp, wl = workload.make("mobile_robot", 1)
    direct = capi.BatchSolver(sp, 1)
    integrator = acados_shim.AcadosSimSolver(None, json_file="mobile_robot_acados_ocp.json")
    rng = np.random.Generator(np.random.PCG64(5))
    x = np.array([0.3, -0.2, 0.7])
    for _ in range(10):
        u = rng.uniform([-0.2, -0.4], [0.35, 0.4])
        # scripts/mpc_acados.py:270-276
        integrator.set("x", x)
        integrator.set("u", u)
        assert integrator.solve() == 0
        xn = integrator.get("x")
        assert np.array_equal(xn, direct.sim_step(x.reshape(1, 3), u.reshape(1, 2))[0])
        assert np.array_equal(integrator.simulate(x, u), xn)
        x = xn
    with pytest.raises(Exception):
        integrator.set("z", np.zeros(1))
    direct.close()
