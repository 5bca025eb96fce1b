"""mpc_amd/pathgen.py against the reference's own data files: its saved
interpolated runs scripts/run1..5.txt and the state_refs / input_refs / kappa it generated from
them (scripts/paths/*.txt, '%.8f' text), copied as fixtures by tests/golden/make_pathgen_fixture.py."""
import os

import numpy as np

from mpc_amd import pathgen
from mpc_amd.optimizer import load_txt

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pathgen")


def test_references_reproduce_the_reference_files():
    runs = [load_txt(os.path.join(G, f"run{i}.txt")) for i in range(1, 6)]
    s, u, k = (load_txt(os.path.join(G, n + ".txt")) for n in ("state_refs", "input_refs", "kappa"))
    # the files were made with v_ref = 1.3 (ramp 0, 0.26, ... 1.3 over int(density) = 6 waypoints),
    # T = 0.125 (waypoint spacing 0.1625 m), N = 40 (454 = 414 filtered waypoints + N + 4)
    r = pathgen.references(runs, 1.3, 40, 0.125)
    assert r["state_refs"].shape == s.shape == (454, 3) and r["input_refs"].shape == u.shape == (455, 2)
    assert np.abs(r["state_refs"][:, :2] - s[:, :2]).max() == 0.0          # positions: exact
    assert np.abs(r["state_refs"][:, 2] - s[:, 2]).max() < 5e-8            # smoothing-spline heading vs %.8f text
    assert np.abs(r["input_refs"] - u).max() < 1e-12
    assert np.abs(r["kappa"] - k.ravel()).max() < 2e-7
    assert int(r["density"]) == 6


def test_resampling_density_and_yaw_unwrap(tmp_path):
    th = np.linspace(0, 1.5 * np.pi, 25)
    run = np.vstack((3 * np.cos(th), 3 * np.sin(th))).T
    rs = pathgen.resample_run(run, 0.5, 0.1)                                # 20 waypoints per metre
    L = 3 * 1.5 * np.pi
    assert abs(len(rs) - L * 20) <= 2
    r = pathgen.references([run], 0.5, 50, 0.1, resample=True)
    yaw = r["state_refs"][:, 2]
    assert np.abs(np.diff(yaw)).max() < 0.1                                  # unwrapped: no 2*pi jumps on a 270 deg arc
    assert abs((yaw[-1] - yaw[0]) - 1.5 * np.pi) < 0.3                    # the s=0.1 smoothing spline bends the ends
    assert np.allclose(r["kappa"][40:-100], 1 / 3, atol=0.07)                # smoothing (s=0.1) ripples around 1/R
    assert r["input_refs"][0, 0] == 0.0 and r["input_refs"][25, 0] == 0.5 and (r["input_refs"][:, 1] == 0).all()
    p = tmp_path / "s.txt"
    pathgen.save_txt(str(p), r["state_refs"])                                # text round trip in the reference's format
    assert np.abs(load_txt(str(p)) - r["state_refs"]).max() <= 5e-9
