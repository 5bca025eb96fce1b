"""The acados-named drop-in layer (include/acados_compat/, mpc_amd/csrc/acados_compat.cpp): the
entry points the reference's callers bind — `<name>_acados_create_capsule/create/solve/reset/
batch_solve/free`, `ocp_nlp_cost_model_set`, `ocp_nlp_constraints_model_set`, `ocp_nlp_out_get/set`,
`sim_in_set/sim_out_get`, `<name>_acados_sim_solve` (reference: src/c_generated_code/
acados_solver_mobile_robot.h:122-161, src/optimizer.cpp:13-119,251-296,323-338).

CPU (`-m "not gpu"`): every declared symbol is exported; the dimension macros agree with the variant
table; the reference's OWN generated example programs (src/c_generated_code*/main_*.c) compile and
link against the compat headers + libmpc_amd.so from where they lie (compile / link only — they are
never run, and only in the build container where /root/reference exists); without a GPU create()
fails instead of falling back.

GPU (`-m gpu`): a plain-C caller (tests/cpp/acados_compat_main.c) that makes the reference's call
sequence, per variant, against the same sequence made through mpc_batch_* (bit for bit) and the CPU
oracle (1e-9 where the HPIPM iteration counts agree, 1e-6 hard)."""
import ctypes
import os
import re
import struct
import subprocess

import numpy as np
import pytest

from mpc_amd import capi, variants, workload

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INC = os.path.join(ROOT, "include", "acados_compat")
REFERENCE = "/root/reference"
REF_DIR = {"mobile_robot": "c_generated_code", "mobile_robot_18": "c_generated_code_18",
           "mobile_robot_25": "c_generated_code_25", "mobile_robot_50": "c_generated_code_50",
           "park": "c_generated_code_park"}


def _strip_comments(text):
    return re.sub(r"/\*.*?\*/", "", text, flags=re.S)


def _declared_symbols():
    names = set()
    for inc in ("mpc_acados_solver_decl.inc", "mpc_acados_sim_decl.inc"):
        text = _strip_comments(open(os.path.join(INC, inc)).read())
        suffixes = re.findall(r"MPC_ACADOS_FN\((_acados_[a-z_A-Z]+)\)\s*\(", text)
        assert len(suffixes) >= 12, inc
        for v in variants.VARIANTS:
            names.update(v + s for s in suffixes)
    for hdr in ("acados_c/ocp_nlp_interface.h", "acados_c/sim_interface.h", "acados/utils/print.h",
                "blasfeo/include/blasfeo_d_aux_ext_dep.h"):
        text = _strip_comments(open(os.path.join(INC, hdr)).read())
        names.update(re.findall(r"\b((?:ocp_nlp|sim_in|sim_out|d_print)[a-z_]*)\s*\(", text))
    return sorted(names)


def test_library_exports_every_acados_named_symbol():
    L = capi.lib()
    names = _declared_symbols()
    assert len(names) == 5 * (22 + 13) + 8 + 2 + 4, len(names)
    for n in names:
        assert hasattr(L, n), f"libmpc_amd.so does not export {n}"
    for must in ("mobile_robot_acados_create_capsule", "mobile_robot_acados_create", "mobile_robot_acados_solve",
                 "mobile_robot_acados_batch_solve", "ocp_nlp_out_get", "ocp_nlp_cost_model_set",
                 "ocp_nlp_constraints_model_set", "sim_in_set", "sim_out_get", "park_acados_reset",
                 "mobile_robot_18_acados_sim_solve"):
        assert must in names


def test_dimension_macros_match_the_variant_table():
    for v in variants.VARIANTS:
        text = open(os.path.join(INC, f"acados_solver_{v}.h")).read()
        m = re.search(rf"#define {v.upper()}_N\s+(\d+)", text)
        assert m and int(m.group(1)) == variants.spec(v).N, v
        for k, val in (("NX", 3), ("NU", 2), ("NBX0", 3), ("NY", 5), ("NYN", 3), ("NP", 0)):
            assert re.search(rf"#define {v.upper()}_{k}\s+{val}\b", text), (v, k)


@pytest.mark.skipif(not os.path.isdir(os.path.join(REFERENCE, "src", "c_generated_code")),
                    reason="reference tree not mounted (GPU box): compile/link check runs in the build container")
def test_reference_example_programs_compile_and_link_against_the_compat_headers(tmp_path):
    """Relink, don't rewrite: the reference's generated main_<name>.c / main_sim_<name>.c, compiled
    from where they lie with -Wall -Werror against include/acados_compat and linked to libmpc_amd.so."""
    lib_dir = os.path.join(ROOT, "mpc_amd")
    n = 0
    for v, d in REF_DIR.items():
        for prog in (f"main_{v}", f"main_sim_{v}"):
            src = os.path.join(REFERENCE, "src", d, prog + ".c")
            assert os.path.exists(src), src
            exe = str(tmp_path / prog)
            cmd = ["gcc", "-std=c99", "-O1", "-Wall", "-Werror", "-I", INC, "-o", exe, src, "-L" + lib_dir, "-lmpc_amd",
                   "-Wl,-rpath-link,/opt/rocm/lib", "-Wl,-rpath," + lib_dir, "-lm"]
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=120)
            assert r.returncode == 0, r.stderr[-3000:]
            und = subprocess.run(["nm", "-u", exe], capture_output=True, text=True).stdout
            assert f"{v}_acados_sim_solve" in und or f"{v}_acados_solve" in und   # bound to OUR library at link time
            n += 1
    assert n == 10


def test_create_fails_loudly_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    L = capi.lib()
    L.mobile_robot_acados_create_capsule.restype = ctypes.c_void_p
    L.mobile_robot_acados_create.argtypes = [ctypes.c_void_p]
    L.mobile_robot_acados_free_capsule.argtypes = [ctypes.c_void_p]
    cap = L.mobile_robot_acados_create_capsule()
    assert cap
    assert L.mobile_robot_acados_create(cap) == 1     # no CPU fallback behind the acados names either
    L.mobile_robot_acados_free_capsule(cap)
    L.park_acados_sim_solver_create_capsule.restype = ctypes.c_void_p
    L.park_acados_sim_create.argtypes = [ctypes.c_void_p]
    L.park_acados_sim_solver_free_capsule.argtypes = [ctypes.c_void_p]
    sc = L.park_acados_sim_solver_create_capsule()
    assert L.park_acados_sim_create(sc) == 1
    L.park_acados_sim_solver_free_capsule(sc)


# --------------------------------------------------------------------------------------------
def _ticks(name, T):
    """T records (x0, yref, yref_e): instance i of the seeded bench workload at tick 0 — distinct
    poses, so the batched section solves T different problems."""
    sp, wl = workload.make(name, T)
    x0, yref, yref_e = wl.tick(0)
    return sp, x0, yref, yref_e


class _Reader:
    def __init__(self, path):
        self.a = np.fromfile(path, dtype=np.float64)
        self.i = 0

    def take(self, n):
        v = self.a[self.i:self.i + n]
        assert len(v) == n, "output file too short"
        self.i += n
        return v

    def traj(self, N):
        return self.take(3 * (N + 1)).reshape(N + 1, 3), self.take(2 * N).reshape(N, 2), self.take(3 * N).reshape(N, 3)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["mobile_robot", "mobile_robot_18", "mobile_robot_25", "mobile_robot_50", "park"])
def test_c_caller_through_the_acados_names_equals_mpc_batch_and_oracle(oracle, name, tmp_path):
    exe = os.path.join(ROOT, "tests", "cpp", "_build", f"acados_compat_{name}")
    assert os.path.exists(exe), "build it with `make -C tests/cpp` (done by __graft_entry__.build())"
    T = 6
    sp, x0, yref, yref_e = _ticks(name, T)
    N = sp.N
    fin, fout = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    with open(fin, "wb") as f:
        f.write(struct.pack("ii", T, N))
        for t in range(T):
            f.write(x0[t].tobytes()); f.write(yref[t].tobytes()); f.write(yref_e[t].tobytes())
    r = subprocess.run([exe, fin, fout], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    rd = _Reader(fout)

    # section 1
    assert list(rd.take(3)) == [N, 3, 2]

    # section 2: T closed ticks on ONE capsule (warm iterate carried inside), + plant step
    one = capi.BatchSolver(sp, 1)
    xo = uo = None
    for t in range(T):
        rg = one.solve(x0[t:t + 1], yref[t:t + 1], yref_e[t:t + 1])
        ro = oracle.rti_batch(sp, x0[t:t + 1], yref[t:t + 1], yref_e[t:t + 1], xo, uo)
        xo, uo = ro["xtraj"], ro["utraj"]
        xn = one.sim_step(x0[t:t + 1], rg["u0"])[0]
        rec = rd.take(8)
        assert int(rec[0]) == rg["status"][0] == ro["status"][0]
        assert np.array_equal(rec[1:3], rg["u0"][0]), (t, rec[1:3], rg["u0"][0])      # bit for bit
        assert np.array_equal(rec[3:6], xn)
        assert int(rec[6]) == rg["qp_iter"][0] and int(rec[7]) == 1
        tol = 1e-9 if rg["qp_iter"][0] == ro["qp_iter"][0] else 1e-6
        assert np.abs(rec[1:3] - ro["u0"][0]).max() < tol
        assert np.abs(rec[3:6] - oracle.sim_step(sp, x0[t:t + 1], ro["u0"])[0]).max() < tol
    # section 3: the iterate after the last tick
    xg, ug, pig = one.get(1)
    xc, uc, pc = rd.traj(N)
    assert np.array_equal(xc, xg[0]) and np.array_equal(uc, ug[0]) and np.array_equal(pc, pig[0])
    assert np.abs(xc - xo[0]).max() < 1e-6 and np.abs(uc - uo[0]).max() < 1e-6

    # section 4: reset -> the cold tick again
    assert int(rd.take(1)[0]) == 0
    one.reset()
    rg = one.solve(x0[0:1], yref[0:1], yref_e[0:1])
    rec = rd.take(3)
    assert int(rec[0]) == 0 and np.array_equal(rec[1:3], rg["u0"][0])
    ro = oracle.rti_batch(sp, x0[0:1], yref[0:1], yref_e[0:1])
    assert np.abs(rec[1:3] - ro["u0"][0]).max() < (1e-9 if rg["qp_iter"][0] == ro["qp_iter"][0] else 1e-6)

    # section 5: ocp_nlp_out_set of the whole trajectory, then a solve
    xi = np.repeat(x0[1:2, None, :], N + 1, axis=1)
    ui = np.tile(np.array([0.1, 0.01]), (1, N, 1))
    assert np.array_equal(rd.take(3), x0[1])
    rg = one.solve(x0[1:2], yref[1:2], yref_e[1:2], xtraj=xi, utraj=ui)
    ro = oracle.rti_batch(sp, x0[1:2], yref[1:2], yref_e[1:2], xi, ui)
    rec = rd.take(3)
    assert int(rec[0]) == rg["status"][0] == ro["status"][0]
    assert np.array_equal(rec[1:3], rg["u0"][0])
    xc, uc, pc = rd.traj(N)
    xg, ug, pig = one.get(1)
    assert np.array_equal(xc, xg[0]) and np.array_equal(uc, ug[0]) and np.array_equal(pc, pig[0])
    tol = 1e-9 if rg["qp_iter"][0] == ro["qp_iter"][0] else 1e-6
    assert np.abs(xc - ro["xtraj"][0]).max() < tol and np.abs(uc - ro["utraj"][0]).max() < tol
    one.close()

    # section 6: batch_solve over T capsules, cold + warm, a single solve on capsule 1, a third batch
    many = capi.BatchSolver(sp, T)
    xo = uo = None
    for rep in range(3):
        if rep == 2:
            xg, ug, _ = many.get(T)
            single = capi.BatchSolver(sp, 1)
            r1 = single.solve(x0[1:2], yref[1:2], yref_e[1:2], xtraj=xg[1:2], utraj=ug[1:2])
            rec = rd.take(3)
            assert int(rec[0]) == r1["status"][0] and np.array_equal(rec[1:3], r1["u0"][0])
            xg[1], ug[1] = r1["xtraj"][0], r1["utraj"][0]
            single.close()
            r1o = oracle.rti_batch(sp, x0[1:2], yref[1:2], yref_e[1:2], xo[1:2], uo[1:2])
            xo[1], uo[1] = r1o["xtraj"][0], r1o["utraj"][0]
            many.close()
            many = capi.BatchSolver(sp, T)   # the capsules moved into a new handle, iterates loaded
            rg = many.solve(x0, yref, yref_e, xtraj=xg, utraj=ug)
        else:
            rg = many.solve(x0, yref, yref_e)
        ro = oracle.rti_batch(sp, x0, yref, yref_e, xo, uo)
        xo, uo = ro["xtraj"], ro["utraj"]
        rec = rd.take(3 * T).reshape(T, 3)
        assert np.array_equal(rec[:, 0].astype(int), rg["status"]) and np.array_equal(rg["status"], ro["status"])
        assert np.array_equal(rec[:, 1:3], rg["u0"]), rep
        same = rg["qp_iter"] == ro["qp_iter"]
        assert np.abs(rec[:, 1:3] - ro["u0"]).max() < 1e-6
        if same.any():
            assert np.abs(rec[same, 1:3] - ro["u0"][same]).max() < 1e-9
    xg, ug, pig = many.get(T)
    xc, uc, pc = rd.traj(N)
    assert np.array_equal(xc, xg[T - 1]) and np.array_equal(uc, ug[T - 1]) and np.array_equal(pc, pig[T - 1])
    many.close()

    # section 7: refused sets return MPC_ACADOS_UNSUPPORTED (-100) and do not exit; benign ones 0
    assert list(rd.take(6)) == [-100, -100, -100, 0, 0, 0]
    assert rd.i == len(rd.a), "unread output"
    assert "qp_iter" in r.stdout          # <name>_acados_print_stats
    assert r.stderr.count("acados_compat") == 3


EXAMPLES = os.path.join(ROOT, "oracle", "_ref", "examples")


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(os.path.join(EXAMPLES, "main_mobile_robot")),
                    reason="oracle/_ref/examples not built (make -C oracle where the reference tree is mounted)")
@pytest.mark.parametrize("name", sorted(REF_DIR))
def test_reference_example_programs_run_on_the_gpu(oracle, name):
    """The reference's OWN generated example programs — src/c_generated_code*/main_<name>.c (create with
    discretisation, x0 through lbx/ubx, initial iterate through ocp_nlp_out_set, rti_phase, solve, time_tot,
    trajectories through ocp_nlp_out_get, print_stats, free) and main_sim_<name>.c (three integrator steps) —
    compiled unmodified in the build container (oracle/Makefile: ref_examples) against include/acados_compat
    and libmpc_amd.so, run here.  They must report success, and the trajectories they print must be the
    oracle's for the same problem (x0 = 0, the generated yref = 0, zero iterate) to the printed precision."""
    sp = variants.spec(name)
    N = sp.N
    r = subprocess.run([os.path.join(EXAMPLES, f"main_{name}")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    out = r.stdout
    assert f"{name}_acados_solve(): SUCCESS!" in out and "SQP iterations  1" in out
    body = out.split("--- xtraj ---")[1]
    xs, us = body.split("--- utraj ---")
    us = us.split("solved ocp")[0]
    xt = np.array([float(t) for t in xs.split()]).reshape(N + 1, 3)
    ut = np.array([float(t) for t in us.split()]).reshape(N, 2)
    ro = oracle.rti_batch(sp, np.zeros((1, 3)), np.zeros((1, N, 5)), np.zeros((1, 3)))
    assert ro["status"][0] == 0
    assert np.abs(xt - ro["xtraj"][0]).max() < 1e-6 and np.abs(ut - ro["utraj"][0]).max() < 1e-6
    m = re.search(r"minimum time for 1 solve ([0-9.]+) \[ms\]", out)
    assert m and 0.0 < float(m.group(1)) < 1000.0
    # the integrator example: three steps from x = 0 with u = 0 stay at 0
    r = subprocess.run([os.path.join(EXAMPLES, f"main_sim_{name}")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    assert "Performed 3 simulation steps with acados integrator successfully." in r.stdout
    vals = [float(t) for blk in r.stdout.split("x_current,")[1:] for t in blk.split("\n")[1:4]]
    assert len(vals) == 9 and max(abs(v) for v in vals) == 0.0

