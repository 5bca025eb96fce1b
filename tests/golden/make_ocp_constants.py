"""Build-container only: extracts the OCP constants the reference bakes into its generated solvers
into tests/golden/ocp_constants.json (a data fixture: numbers, no source text).

Two independent reference-held sources per variant:
  * the literals of src/c_generated_code*/acados_solver_<name>.c / .h  (what the deployed solver runs)
  * the JSON dump scripts/<name>_acados_ocp.json                       (what the Python codegen wrote)
They disagree for `mobile_robot_50` (N = 80 in src, 50 in the JSON): both are kept,
the product follows src/ (what CMake builds).

    python tests/golden/make_ocp_constants.py   # needs /root/reference
"""
import json
import os
import re
import sys

REF = os.environ.get("MPC_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))

VARIANTS = {  # name -> (generated-code dir, symbol prefix)
    "mobile_robot": ("src/c_generated_code", "mobile_robot"),
    "mobile_robot_18": ("src/c_generated_code_18", "mobile_robot_18"),
    "mobile_robot_25": ("src/c_generated_code_25", "mobile_robot_25"),
    "mobile_robot_50": ("src/c_generated_code_50", "mobile_robot_50"),
    "park": ("src/c_generated_code_park", "park"),
}


def num(s):
    return float(s)


def from_c(name, cdir, prefix):
    c = open(os.path.join(REF, cdir, f"acados_solver_{prefix}.c")).read()
    h = open(os.path.join(REF, cdir, f"acados_solver_{prefix}.h")).read()
    out = {}
    out["N"] = int(re.search(r"#define\s+%s_N\s+(\d+)" % prefix.upper(), h).group(1))
    for k in ("NX", "NU", "NY", "NYN", "NBX", "NBX0", "NBU", "NBXN"):
        m = re.search(r"#define\s+%s_%s\s+(\d+)" % (prefix.upper(), k), h)
        out[k] = int(m.group(1)) if m else None

    def diag(var, dim):
        vals = {}
        for m in re.finditer(r"\b%s\[(\d+)\+\(%s\) \* (\d+)\] = ([-+0-9.eE]+);" % (var, dim), c):
            i, j, v = int(m.group(1)), int(m.group(2)), num(m.group(3))
            assert i == j, f"{name}: off-diagonal weight {var}[{i},{j}]"
            vals[i] = v
        return [vals[i] for i in sorted(vals)]

    out["W"] = diag("W", "NY")
    out["W_e"] = diag("W_e", "NYN")

    def vec(var):
        vals = {}
        for m in re.finditer(r"^\s*%s\[(\d+)\] = ([-+0-9.eE]+);" % var, c, flags=re.M):
            vals[int(m.group(1))] = num(m.group(2))
        return [vals[i] for i in sorted(vals)]

    for v in ("lbu", "ubu", "lbx", "ubx", "idxbu", "idxbx", "idxbx0", "idxbxe_0"):
        out[v] = vec(v)
    # time step and cost scaling: either `double time_step = X;` (uniform) or a time_steps array
    m = re.search(r"double time_step = ([-+0-9.eE]+);", c)
    out["time_step"] = num(m.group(1)) if m else None
    # solver plan and options (string-valued settings as written in the generated code)
    plan = {}
    for key in ("nlp_solver", "ocp_qp_solver_plan.qp_solver", "regularization"):
        m = re.search(r"nlp_solver_plan->%s = (\w+);" % re.escape(key), c)
        plan[key] = m.group(1) if m else None
    plan["nlp_cost"] = sorted(set(re.findall(r"nlp_solver_plan->nlp_cost\[\w+\] = (\w+);", c)))
    plan["nlp_dynamics"] = sorted(set(re.findall(r"nlp_solver_plan->nlp_dynamics\[\w+\] = (\w+);", c)))
    plan["sim_solver"] = sorted(set(re.findall(r"nlp_solver_plan->sim_solver_plan\[\w+\]\.sim_solver = (\w+);", c)))
    plan["nlp_constraints"] = sorted(set(re.findall(r"nlp_solver_plan->nlp_constraints\[\w+\] = (\w+);", c)))
    out["plan"] = plan
    opts = {}
    for key, pat in {
        "sim_method_num_stages": r"int sim_method_num_stages = (\d+);",
        "sim_method_num_steps": r"int sim_method_num_steps = (\d+);",
        "qp_solver_iter_max": r"int qp_solver_iter_max = (\d+);",
        "qp_solver_cond_N_is_N": r"qp_solver_cond_N = (N);",
        "hpipm_mode": r'"qp_hpipm_mode", "(\w+)"',
        "levenberg_marquardt": r"double levenberg_marquardt = ([-+0-9.eE]+);",
        "nlp_solver_step_length": r"double nlp_solver_step_length = ([-+0-9.eE]+);",
        "full_step_dual": r"int full_step_dual = (\d+);",
        "qp_solver_ric_alg": r"int qp_solver_ric_alg = (\d+);",
        "qp_solver_cond_ric_alg": r"int qp_solver_cond_ric_alg = (\d+);",
        "as_rti_level": r"int as_rti_level = (\d+);",
        "as_rti_iter": r"int as_rti_iter = (\d+);",
    }.items():
        m = re.search(pat, c)
        if m is None:
            opts[key] = None
        else:
            g = m.group(1)
            try:
                opts[key] = float(g) if ("." in g or "e" in g.lower()) and g != "N" else (g if not g.isdigit() else int(g))
            except ValueError:
                opts[key] = g
    out["opts"] = opts
    return out


def from_json(name):
    p = os.path.join(REF, "scripts", f"{name}_acados_ocp.json")
    if not os.path.exists(p):
        return None
    d = json.load(open(p))
    so = d["solver_options"]
    keep = ("Tsim", "tf", "nlp_solver_type", "qp_solver", "qp_solver_iter_max", "qp_solver_cond_N", "qp_solver_warm_start",
            "hpipm_mode", "integrator_type", "hessian_approx", "regularize_method", "levenberg_marquardt",
            "nlp_solver_step_length", "nlp_solver_max_iter", "nlp_solver_tol_stat", "nlp_solver_tol_eq",
            "nlp_solver_tol_ineq", "nlp_solver_tol_comp", "qp_solver_tol_stat", "qp_solver_tol_eq", "qp_solver_tol_ineq",
            "qp_solver_tol_comp", "globalization", "full_step_dual", "cost_discretization", "as_rti_level", "as_rti_iter",
            "num_threads_in_batch_solve")
    ts = so.get("time_steps") or []
    return {
        "N": d["dims"]["N"], "nx": d["dims"]["nx"], "nu": d["dims"]["nu"], "ny": d["dims"]["ny"], "ny_e": d["dims"]["ny_e"],
        "nbx": d["dims"]["nbx"], "nbu": d["dims"]["nbu"], "nbx_0": d["dims"]["nbx_0"], "nbxe_0": d["dims"].get("nbxe_0"),
        "nbx_e": d["dims"]["nbx_e"],
        "W_diag": [d["cost"]["W"][i][i] for i in range(len(d["cost"]["W"]))],
        "W_offdiag_max": max(abs(d["cost"]["W"][i][j]) for i in range(5) for j in range(5) if i != j),
        "W_e_diag": [d["cost"]["W_e"][i][i] for i in range(len(d["cost"]["W_e"]))],
        "Vx": d["cost"]["Vx"], "Vu": d["cost"]["Vu"], "Vx_e": d["cost"]["Vx_e"],
        "cost_type": d["cost"]["cost_type"], "cost_type_e": d["cost"]["cost_type_e"],
        "lbu": d["constraints"]["lbu"], "ubu": d["constraints"]["ubu"], "lbx": d["constraints"]["lbx"],
        "ubx": d["constraints"]["ubx"], "idxbu": d["constraints"]["idxbu"], "idxbx": d["constraints"]["idxbx"],
        "idxbxe_0": d["constraints"].get("idxbxe_0"),
        "time_step_min": min(ts) if ts else None, "time_step_max": max(ts) if ts else None,
        "sim_method_num_stages": sorted(set(so["sim_method_num_stages"])) if isinstance(so["sim_method_num_stages"], list) else so["sim_method_num_stages"],
        "sim_method_num_steps": sorted(set(so["sim_method_num_steps"])) if isinstance(so["sim_method_num_steps"], list) else so["sim_method_num_steps"],
        "solver_options": {k: so.get(k) for k in keep},
    }


def main():
    if not os.path.isdir(os.path.join(REF, "src")):
        sys.exit(f"{REF} is not mounted: this script only runs in the build container")
    out = {"_source": "reference src/c_generated_code*/acados_solver_*.{c,h} and scripts/*_acados_ocp.json "
                      "(numbers only; made by tests/golden/make_ocp_constants.py)"}
    for name, (cdir, prefix) in VARIANTS.items():
        out[name] = {"generated_c": from_c(name, cdir, prefix), "json": from_json(name)}
    with open(os.path.join(HERE, "ocp_constants.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote", os.path.join(HERE, "ocp_constants.json"))


if __name__ == "__main__":
    main()
