"""world_size-2 gloo test of the N>1 path on CPU: every rank solves its contiguous shard with no
data-path collective; the optional final gather reassembles u0 in instance order and equals the
single-process result bit for bit.  (The per-rank solver stand-in is the kernel algorithm
compiled for the host; on the GPU box the same sharding code drives libmpc_amd.so.)"""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, scaling, batch, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    from mpc_amd import sharding, workload
    from tests import hostcore_lib
    # the same calls bench.py makes: connect (nccl would fall back to gloo; here gloo directly),
    # a contiguous shard per rank, timed_steps (warm-up, barrier-bracketed timed region, MAX over ranks)
    ranks = sharding.connect("gloo", rank, world)
    assert ranks.world == world and ranks.backend == "gloo"
    lo, hi, total = sharding.rank_slice(scaling, batch, world, rank)   # bench.py's own call, both of its modes
    sp, wl = workload.make("mobile_robot_18", hi - lo, first=lo)
    state = {"x": None, "u": None, "r": None, "seen": []}

    def step(t):
        x0, yref, yref_e = wl.tick(t)
        state["r"] = hostcore_lib.rti_batch(sp, x0, yref, yref_e, state["x"], state["u"])
        state["x"], state["u"] = state["r"]["xtraj"], state["r"]["utraj"]

    elapsed = sharding.timed_steps(step, lambda: None, 1, 2, ranks, after_step=state["seen"].append)
    assert state["seen"] == [1, 2] and elapsed > 0.0
    t = ranks.max(1.0 + rank)
    u0_all = sharding.gather_rows(state["r"]["u0"], total)
    np.save(os.path.join(out_dir, f"u0_rank{rank}.npy"), u0_all)
    np.save(os.path.join(out_dir, f"t_rank{rank}.npy"), np.array([t, elapsed]))
    ranks.close()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("scaling,batch,total", [("strong", 37, 37), ("weak", 18, 36)])
def test_two_rank_sharded_solve_equals_single_process(tmp_path, scaling, batch, total):
    """strong: ONE batch of 37 cut into ragged shards of 18 and 19 (BASELINE.json configs[2]);
    weak: 18 instances per rank, global batch 36 (bench.py's default)."""
    import torch.multiprocessing as mp
    from mpc_amd import sharding, workload
    from tests import hostcore_lib
    world = 2
    assert [sharding.rank_slice(scaling, batch, world, r)[2] for r in range(world)] == [total, total]
    port = _free_port()
    mp.spawn(_worker, args=(world, port, scaling, batch, str(tmp_path)), nprocs=world, join=True)
    sp, wl = workload.make("mobile_robot_18", total)
    xo = uo = None
    for t in range(3):   # warm-up tick + two timed ticks, as in the workers
        x0, yref, yref_e = wl.tick(t)
        ref = hostcore_lib.rti_batch(sp, x0, yref, yref_e, xo, uo)
        xo, uo = ref["xtraj"], ref["utraj"]
    times = []
    for r in range(world):
        got = np.load(tmp_path / f"u0_rank{r}.npy")
        assert got.shape == (total, 2) and np.array_equal(got, ref["u0"])
        tt = np.load(tmp_path / f"t_rank{r}.npy")
        assert tt[0] == float(world)
        times.append(tt[1])
    assert times[0] == times[1]          # the timed region's wall time is the MAX over ranks on every rank


def _fallback_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    from mpc_amd import sharding
    msgs = []
    # what bench.py does on a node where RCCL cannot start: "nccl" fails on every rank (here: no
    # GPU at all), connect() opens a fresh gloo rendezvous on the next port hosted by rank 0
    ranks = sharding.connect("nccl", rank, world, device=None, log=msgs.append)
    assert ranks.backend == "gloo" and ranks.world == world and msgs and "gloo" in msgs[0]
    ranks.barrier()
    t = ranks.max(10.0 * (rank + 1))
    np.save(os.path.join(out_dir, f"fb_rank{rank}.npy"), np.array([t]))
    ranks.close()


@pytest.mark.timeout(300)
def test_rccl_start_failure_falls_back_to_gloo(tmp_path):
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        pytest.skip("RCCL can start here: nothing to fall back from")
    import torch.multiprocessing as mp
    world = 2
    mp.spawn(_fallback_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert np.load(tmp_path / f"fb_rank{r}.npy")[0] == 10.0 * world
