"""mpc_batch_solve_static — the entry point that allocates nothing (capsules in a file-scope
device array of the library) — against mpc_batch_solve_device on a handle, through the C-ABI:
same kernels, so the outputs are compared bit for bit (the bar is north_star's 1e-6).  Two calls
in a row, the second with a smaller batch; cold and warm semantics; size limits."""
import os

import numpy as np
import pytest

from mpc_amd import capi, workload

pytestmark = pytest.mark.gpu


def _dev(torch, a):
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")


def _outs(torch, B):
    return (torch.zeros(B, 2, dtype=torch.float64, device="cuda:0"),
            torch.full((B,), -7, dtype=torch.int32, device="cuda:0"),
            torch.full((B,), -7, dtype=torch.int32, device="cuda:0"))


@pytest.mark.parametrize("name,N", [("mobile_robot", None), ("mobile_robot_25", 25), ("mobile_robot_50", None)])
def test_static_entry_point_matches_the_handle(name, N):
    torch = pytest.importorskip("torch")
    B1, B2 = 300, 120
    kw = {} if N is None else {"N": N}
    sp, wl = workload.make(name, B1, **kw)
    handle = capi.BatchSolver(sp, B1, devices=[0])

    def run_handle(B, t):
        x0, yref, yref_e = wl.tick(t)
        d = [_dev(torch, a[:B]) for a in (x0, yref, yref_e)]
        u0, st, it = _outs(torch, B)
        torch.cuda.synchronize()
        handle.solve_device(B, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), u0.data_ptr(), 0, 0, st.data_ptr(), it.data_ptr())
        handle.last_kernel_ms()
        torch.cuda.synchronize()
        return u0.cpu().numpy(), st.cpu().numpy(), it.cpu().numpy()

    def run_static(B, t, cold):
        x0, yref, yref_e = wl.tick(t)
        d = [_dev(torch, a[:B]) for a in (x0, yref, yref_e)]
        u0, st, it = _outs(torch, B)
        torch.cuda.synchronize()
        rc = capi.solve_static(sp, B, cold, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), u0.data_ptr(), st.data_ptr(), it.data_ptr())
        assert rc == 0
        torch.cuda.synchronize()
        return u0.cpu().numpy(), st.cpu().numpy(), it.cpu().numpy()

    # call 1: cold, B1 instances = a fresh handle
    h1, s1 = run_handle(B1, 0), run_static(B1, 0, True)
    # call 2, straight after, SMALLER batch, warm: continues from call 1's iterates of the first B2 capsules
    h2, s2 = run_handle(B2, 1), run_static(B2, 1, False)
    for (hu, hs, hi), (su, ss, si) in ((h1, s1), (h2, s2)):
        assert (hs == 0).all()
        assert np.array_equal(hs, ss) and np.array_equal(hi, si)
        assert np.abs(hu - su).max() < 1e-6 and np.array_equal(hu, su)
    assert not np.array_equal(h2[0], h1[0][:B2])
    # a cold call clears what the previous calls left (smaller batch again): equals a reset handle
    handle.reset()
    h3, s3 = run_handle(50, 2), run_static(50, 2, True)
    assert np.array_equal(h3[0], s3[0]) and np.array_equal(h3[2], s3[2])
    # outputs that are not asked for, null stream
    x0, yref, yref_e = wl.tick(3)
    d = [_dev(torch, a[:50]) for a in (x0, yref, yref_e)]
    u0 = torch.zeros(50, 2, dtype=torch.float64, device="cuda:0")
    torch.cuda.synchronize()
    assert capi.solve_static(sp, 50, False, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), u0.data_ptr()) == 0
    torch.cuda.synchronize()
    h4 = run_handle(50, 3)
    assert np.array_equal(h4[0], u0.cpu().numpy())
    handle.close()


def test_static_entry_point_limits():
    torch = pytest.importorskip("torch")
    sp, wl = workload.make("mobile_robot", 4)
    x0, yref, yref_e = wl.tick(0)
    d = [_dev(torch, a) for a in (x0, yref, yref_e)]
    u0 = torch.zeros(4, 2, dtype=torch.float64, device="cuda:0")
    args = (d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), u0.data_ptr())
    too_many = capi.STATIC_MAX_LINES // (sp.N + 1) + 1
    assert capi.solve_static(sp, too_many, True, *args) == -4          # refused before anything is read
    assert capi.solve_static(sp, capi.STATIC_MAX_B + 1, True, *args) == -4
    assert capi.solve_static(sp, 0, True, *args) == -1
    assert capi.solve_static(sp.copy(sqp_max_iter=3), 4, True, *args) == -1
    assert capi.solve_static(sp, 4, True, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), 0) == -1   # u0 is required
    assert capi.solve_static(sp, 4, True, *args) == 0
    torch.cuda.synchronize()
    # the largest batch the workspace takes at this horizon runs and equals a handle on a sample
    Bmax = capi.STATIC_MAX_LINES // (sp.N + 1)
    spb, wlb = workload.make("mobile_robot", Bmax)
    xb, yb, eb = wlb.tick(0)
    db = [_dev(torch, a) for a in (xb, yb, eb)]
    ub = torch.zeros(Bmax, 2, dtype=torch.float64, device="cuda:0")
    torch.cuda.synchronize()
    assert capi.solve_static(spb, Bmax, True, db[0].data_ptr(), db[1].data_ptr(), db[2].data_ptr(), ub.data_ptr()) == 0
    torch.cuda.synchronize()
    h = capi.BatchSolver(spb, Bmax, devices=[0])
    r = h.solve(xb, yb, eb)
    assert np.array_equal(r["u0"], ub.cpu().numpy())
    h.close()


def test_static_warm_call_must_match_what_the_array_holds():
    """cold = 0 continues the iterates the previous call left: same horizon, and only capsules an
    earlier call has initialised — anything else is refused with -1 instead of reading capsule
    lines of another layout."""
    torch = pytest.importorskip("torch")
    sp, wl = workload.make("mobile_robot", 8)
    x0, yref, yref_e = wl.tick(0)
    d = [_dev(torch, a) for a in (x0, yref, yref_e)]
    u0 = torch.zeros(8, 2, dtype=torch.float64, device="cuda:0")
    sp25, wl25 = workload.make("mobile_robot_25", 8, N=25)
    a25 = [_dev(torch, a) for a in wl25.tick(0)]
    a = (d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), u0.data_ptr())
    b = (a25[0].data_ptr(), a25[1].data_ptr(), a25[2].data_ptr(), u0.data_ptr())
    # (the array is shared by every test of this process: a cold call at another horizon starts the count afresh)
    assert capi.solve_static(sp25, 4, True, *b) == 0
    assert capi.solve_static(sp25, 4, False, *b) == 0
    assert capi.solve_static(sp25, 8, False, *b) == -1   # capsules 4..7 never initialised at this horizon
    assert capi.solve_static(sp, 4, False, *a) == -1     # other horizon: capsule lines of another layout
    assert capi.solve_static(sp, 4, True, *a) == 0
    assert capi.solve_static(sp, 4, False, *a) == 0
    assert capi.solve_static(sp, 8, False, *a) == -1
    assert capi.solve_static(sp25, 4, False, *b) == -1
    torch.cuda.synchronize()


def test_static_call_captured_in_a_graph_replays_like_a_handle():
    """The use the entry point exists for: ONE warm call recorded in a HIP graph and replayed.  The
    workload (cost weights x 1e4, +-0.3 m state box) makes the scan kernel decline about one instance
    in ten, so every replay goes through the list of declined instances and the sequential kernel
    behind it; which of the two list counters a solve uses is device state (Mem::REDO), so a replay
    counts from zero like a fresh call.  Against a handle making the same calls: bit for bit, three
    replays in a row.  (The handle is created with MPC_RESUME=0: like the static array it then has no
    records for the interior-point iterates of declined instances, and its list kernel solves them from
    the cold start as the entry point's does.)"""
    torch = pytest.importorskip("torch")
    B = 4000   # 2000 sets of two instances: more than the chip's 1024 wavefront slots, i.e. the persistent scan kernel,
    #            whose work counter must put itself back to zero for the next replay
    sp, wl = workload.make("mobile_robot", B)
    x0, yref, yref_e = (a.copy() for a in wl.tick(0))
    shift = x0[:, :2].copy()
    x0[:, :2] -= shift; yref[:, :, :2] -= shift[:, None, :]; yref_e[:, :2] -= shift
    sp = sp.copy(lbx=(-0.3, -0.3), ubx=(0.3, 0.3), W=tuple(1e4 * w for w in sp.W), W_e=tuple(1e4 * w for w in sp.W_e))
    d = [_dev(torch, a) for a in (x0, yref, yref_e)]
    keep = os.environ.get("MPC_RESUME")
    os.environ["MPC_RESUME"] = "0"
    try:
        handle = capi.BatchSolver(sp, B, devices=[0])
    finally:
        if keep is None:
            del os.environ["MPC_RESUME"]
        else:
            os.environ["MPC_RESUME"] = keep
    ref, declined = [], 0
    for r in range(4):
        u0, st, it = _outs(torch, B)
        torch.cuda.synchronize()
        handle.solve_device(B, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), u0.data_ptr(), 0, 0, st.data_ptr(), it.data_ptr())
        handle.last_kernel_ms()
        torch.cuda.synchronize()
        declined += int(((handle.get_diag(B) & 4) != 0).sum())
        ref.append((u0.cpu().numpy(), st.cpu().numpy(), it.cpu().numpy()))
    assert declined > 4 * 0.02 * B, "the workload no longer makes the scan kernel decline anything"
    u0, st, it = _outs(torch, B)
    torch.cuda.synchronize()
    # call 0: cold, eager (also looks up the array and loads the kernels before the capture starts)
    assert capi.solve_static(sp, B, True, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), u0.data_ptr(), st.data_ptr(), it.data_ptr()) == 0
    torch.cuda.synchronize()
    assert np.array_equal(u0.cpu().numpy(), ref[0][0]) and np.array_equal(it.cpu().numpy(), ref[0][2])
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        rc = capi.solve_static(sp, B, False, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), u0.data_ptr(), st.data_ptr(), it.data_ptr(),
                               stream=torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    for r in range(1, 4):
        st.fill_(-7); it.fill_(-7); u0.zero_()
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        assert (st.cpu().numpy() >= 0).all(), r            # no PENDING status left behind
        assert np.array_equal(st.cpu().numpy(), ref[r][1]) and np.array_equal(it.cpu().numpy(), ref[r][2]), r
        assert np.array_equal(u0.cpu().numpy(), ref[r][0]), r
    handle.close()
