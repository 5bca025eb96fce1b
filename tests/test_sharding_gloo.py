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


def _worker(rank, world, port, total, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    from mpc_amd import sharding, workload
    from tests import hostcore_lib
    sharding.init("gloo", rank, world)
    lo, hi = sharding.shard_range(total, world, rank)
    sp, wl = workload.make("mobile_robot_18", hi - lo, first=lo)
    x0, yref, yref_e = wl.tick(0)
    dist.barrier()
    r = hostcore_lib.rti_batch(sp, x0, yref, yref_e)
    dist.barrier()
    t = sharding.max_over_ranks(1.0 + rank)
    u0_all = sharding.gather_rows(r["u0"], total)
    np.save(os.path.join(out_dir, f"u0_rank{rank}.npy"), u0_all)
    np.save(os.path.join(out_dir, f"t_rank{rank}.npy"), np.array([t]))
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_sharded_solve_equals_single_process(tmp_path):
    import torch.multiprocessing as mp
    from mpc_amd import workload
    from tests import hostcore_lib
    total, world = 37, 2   # ragged on purpose: shards of 18 and 19
    port = _free_port()
    mp.spawn(_worker, args=(world, port, total, str(tmp_path)), nprocs=world, join=True)
    sp, wl = workload.make("mobile_robot_18", total)
    x0, yref, yref_e = wl.tick(0)
    ref = hostcore_lib.rti_batch(sp, x0, yref, yref_e)
    for r in range(world):
        got = np.load(tmp_path / f"u0_rank{r}.npy")
        assert got.shape == (total, 2) and np.array_equal(got, ref["u0"])
        assert np.load(tmp_path / f"t_rank{r}.npy")[0] == float(world)
