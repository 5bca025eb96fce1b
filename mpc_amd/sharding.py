"""Multi-GPU sharding of a batch of independent OCP instances.

Instances have no coupling term, so a batch is cut into contiguous slices
[g*B/G, (g+1)*B/G), one per rank/GPU, and every rank solves its slice with no collective on
the data path.  torch.distributed (backend "nccl" = RCCL on ROCm, "gloo" in CPU tests) is used
only for the barrier around the timed region, the max-over-ranks time and the optional final
gather of u0 (B x 2 doubles).  The reference has no counterpart: its batch entry point is a
serial loop (src/c_generated_code/acados_solver_mobile_robot.c:951-961).
"""
import os

import numpy as np


def shard_range(total, world, rank):
    """Contiguous slice of `total` instances owned by `rank` (sizes differ by at most 1)."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    lo = (total * rank) // world
    hi = (total * (rank + 1)) // world
    return lo, hi


def rank_slice(scaling, batch, world, rank):
    """This rank's contiguous slice [lo, hi) of the global batch and the global batch size.
    "weak": every rank owns `batch` instances (the global batch grows with the world size);
    "strong": ONE batch of `batch` instances is cut over the ranks (BASELINE.json configs[2]:
    "batch 65536 sharded across 1/2/4/8")."""
    if scaling == "weak":
        lo, hi = shard_range(world * batch, world, rank)
        return lo, hi, world * batch
    if scaling == "strong":
        lo, hi = shard_range(batch, world, rank)
        return lo, hi, batch
    raise ValueError("scaling must be 'weak' or 'strong'")


def env_rank():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init(backend, rank, world, device=None):
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    kw = {}
    if device is not None and backend == "nccl":
        kw["device_id"] = device
    dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return dist


def max_over_ranks(value, device="cpu"):
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_rows(local, total, device="cpu"):
    """Optional final gather: concatenates per-rank row blocks (e.g. u0 [n_g, 2]) in rank
    order on every rank.  Never on the critical path."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return np.asarray(local)
    world = dist.get_world_size()
    width = local.shape[1]
    sizes = [shard_range(total, world, r) for r in range(world)]
    nmax = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros(nmax, width, dtype=torch.float64, device=device)
    pad[: local.shape[0]] = torch.as_tensor(local, dtype=torch.float64, device=device)
    outs = [torch.zeros_like(pad) for _ in range(world)]
    dist.all_gather(outs, pad)
    return np.concatenate([outs[r][: hi - lo].cpu().numpy() for r, (lo, hi) in enumerate(sizes)], axis=0)


class Ranks:
    """The distributed context of one benchmark / driver process: rank, world size, the
    torch.distributed group (None for a single process) and the device the scalar collectives
    run on.  The data path never uses it — only `barrier` and `max_over_ranks`."""

    def __init__(self, rank=0, world=1, dist=None, backend=None, device="cpu"):
        self.rank, self.world, self.dist, self.backend, self.device = rank, world, dist, backend, device

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()

    def max(self, value):
        return max_over_ranks(value, self.device) if self.dist is not None else float(value)

    def close(self):
        if self.dist is not None:
            self.dist.barrier()
            self.dist.destroy_process_group()
            self.dist = None


def connect(backend, rank, world, device=None, log=None):
    """init() with the fallback the bench needs: if RCCL ("nccl") cannot start on this node the
    two scalar collectives go over gloo instead (and the result says so)."""
    if world <= 1:
        return Ranks()
    import torch.distributed as dist
    if backend == "nccl":
        try:
            init("nccl", rank, world, device)
            dist.barrier()
            return Ranks(rank, world, dist, "rccl", device)
        except Exception as e:  # noqa: BLE001
            if log:
                log(f"rank {rank}: nccl/RCCL unavailable ({e}); using gloo for barrier/all_reduce")
            if dist.is_initialized():
                dist.destroy_process_group()
            # a fresh rendezvous on the next port, hosted by rank 0 itself (under torchrun the
            # agent hosts the store of MASTER_PORT only, and workers would wait for a server
            # that nobody starts)
            os.environ["MASTER_PORT"] = str(int(os.environ.get("MASTER_PORT", "29500")) + 1)
            os.environ["TORCHELASTIC_USE_AGENT_STORE"] = "False"
            backend = "gloo"
    init(backend, rank, world)
    return Ranks(rank, world, dist, backend, "cpu")


def timed_steps(step, sync, warmup, steps, ranks, after_step=None):
    """The timing protocol of the bench contract: `warmup` untimed steps, then exactly `steps`
    steps bracketed by (device sync + barrier) on both sides; returns the MAX over ranks of the
    wall time in seconds.  step(t) enqueues step t (t counts from 0 through the warm-up),
    sync() waits for this rank's device, after_step(t) may record per-step data."""
    import time
    for t in range(warmup):
        step(t)
    sync()
    ranks.barrier()
    sync()
    t0 = time.perf_counter()
    for t in range(warmup, warmup + steps):
        step(t)
        if after_step:
            after_step(t)
    sync()
    ranks.barrier()
    return ranks.max(time.perf_counter() - t0)
