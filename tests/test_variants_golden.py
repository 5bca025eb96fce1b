"""H3: the constants `<name>_acados_create` bakes into a capsule, pinned by the reference's own
files.  tests/golden/ocp_constants.json was extracted (numbers only) from
src/c_generated_code*/acados_solver_*.{c,h} and scripts/*_acados_ocp.json by
tests/golden/make_ocp_constants.py in the build container; here the product's variant table
(C: mpc_spec_variant; Python: mpc_amd/variants.py) and its fixed algorithmic choices are compared
with it."""
import json
import os

import pytest

from mpc_amd import capi, variants

GOLD = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ocp_constants.json")))
NAMES = [k for k in GOLD if not k.startswith("_")]


@pytest.mark.parametrize("name", NAMES)
def test_variant_table_matches_generated_solver(name):
    g = GOLD[name]["generated_c"]
    for sp in (capi.spec_variant(name), variants.spec(name)):
        assert sp.N == g["N"]
        assert sp.Ts == g["time_step"]                      # ERK step = cost scaling of stages 0..N-1
        assert list(sp.W) == g["W"] and list(sp.W_e) == g["W_e"]
        assert list(sp.lbu) == g["lbu"] and list(sp.ubu) == g["ubu"]
        assert list(sp.lbx) == g["lbx"] and list(sp.ubx) == g["ubx"]
        assert sp.qp_iter_max == g["opts"]["qp_solver_iter_max"] == 50
        assert sp.sqp_max_iter == 1                         # SQP_RTI
        assert sp.hpipm_mode == variants.HPIPM_BALANCE
    # dimensions and index sets the kernels hard-code
    assert (g["NX"], g["NU"], g["NY"], g["NYN"]) == (3, 2, 5, 3)
    assert (g["NBU"], g["NBX"], g["NBX0"], g["NBXN"]) == (2, 2, 3, 0)
    assert g["idxbu"] == [0, 1] and g["idxbx"] == [0, 1] and g["idxbxe_0"] == [0, 1, 2]
    # algorithm selection (acados_solver_<name>.c plan + options)
    p, o = g["plan"], g["opts"]
    assert p["nlp_solver"] == "SQP_RTI" and p["ocp_qp_solver_plan.qp_solver"] == "PARTIAL_CONDENSING_HPIPM"
    assert p["nlp_cost"] == ["LINEAR_LS"] and p["nlp_constraints"] == ["BGH"]
    assert p["nlp_dynamics"] == ["CONTINUOUS_MODEL"] and p["sim_solver"] == ["ERK"]
    assert o["sim_method_num_stages"] == 4 and o["sim_method_num_steps"] == 1     # classical RK4, one step
    assert o["hpipm_mode"] == "BALANCE" and o["qp_solver_cond_N_is_N"] == "N"     # no condensing
    assert o["levenberg_marquardt"] == 0 and o["nlp_solver_step_length"] == 1 and o["full_step_dual"] == 0
    if o["as_rti_level"] is not None:      # newer template only
        assert o["as_rti_level"] == 4 and o["as_rti_iter"] == 1                   # standard RTI


@pytest.mark.parametrize("name", NAMES)
def test_json_dump_agrees_on_structure(name):
    """The Python codegen's JSON dumps are older snapshots for some variants (bounds / N of
    mobile_robot_25 / _50 differ from src/), but structure and solver options agree."""
    j = GOLD[name]["json"]
    assert j is not None
    assert (j["nx"], j["nu"], j["ny"], j["ny_e"], j["nbu"], j["nbx"], j["nbx_0"], j["nbx_e"]) == (3, 2, 5, 3, 2, 2, 3, 0)
    assert j["cost_type"] == j["cost_type_e"] == "LINEAR_LS" and j["W_offdiag_max"] == 0.0
    assert j["Vx"] == [[1, 0, 0], [0, 1, 0], [0, 0, 1], [0, 0, 0], [0, 0, 0]] and j["Vu"] == [[0, 0]] * 3 + [[1, 0], [0, 1]]
    so = j["solver_options"]
    assert so["nlp_solver_type"] == "SQP_RTI" and so["qp_solver"] == "PARTIAL_CONDENSING_HPIPM"
    assert so["hpipm_mode"] == "BALANCE" and so["qp_solver_iter_max"] == 50 and so["qp_solver_warm_start"] == 0
    assert so["integrator_type"] == "ERK" and so["hessian_approx"] == "GAUSS_NEWTON"
    assert j["sim_method_num_stages"] == [4] and j["sim_method_num_steps"] == [1]
    assert so["qp_solver_tol_stat"] is None and so["qp_solver_tol_eq"] is None      # HPIPM mode defaults stay
    assert so["nlp_solver_max_iter"] == 100 and so["nlp_solver_tol_stat"] == 1e-6   # config 5 (full SQP) settings
    assert j["time_step_min"] == pytest.approx(j["time_step_max"], rel=1e-12)
    if name in ("mobile_robot", "mobile_robot_18", "park"):   # dumps that are current
        g = GOLD[name]["generated_c"]
        assert j["N"] == g["N"] and j["W_diag"] == g["W"] and j["W_e_diag"] == g["W_e"]
        assert j["lbu"] == g["lbu"] and j["ubu"] == g["ubu"] and j["lbx"] == g["lbx"] and j["ubx"] == g["ubx"]
        assert j["time_step_min"] == pytest.approx(g["time_step"], rel=1e-12)


def test_unimplemented_modes_and_bad_ranges_are_rejected_by_the_spec_check():
    """spec_problem (mpc_amd/csrc/consts.hpp) runs before any device call: -1 even without a GPU."""
    import ctypes
    L = capi.lib()
    base = variants.spec("mobile_robot")
    bad = [base.copy(hpipm_mode=variants.HPIPM_SPEED), base.copy(hpipm_mode=variants.HPIPM_ROBUST),
           base.copy(hpipm_mode=variants.HPIPM_SPEED_ABS), base.copy(hpipm_mode=7), base.copy(N=variants.MAX_N + 1),
           base.copy(L=-0.27), base.copy(Ts=0.0), base.copy(sqp_max_iter=5, nlp_tol=0.0), base.copy(sqp_max_iter=5, nlp_tol=float("nan"))]
    for sp in bad:
        h = ctypes.c_void_p()
        assert L.mpc_batch_create(ctypes.byref(h), ctypes.byref(sp), 4, None, 0) == -1 and not h
