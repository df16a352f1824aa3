"""Multi-GPU reconstruction: one process per GPU, replicated driver, sharded refinement.

The sharding, the replicate-thin-batches rule and the per-batch all-gather live UNDER the C ABI
(include/pais_mvs.h: pais_mvs_comm_init_rccl / pais_mvs_comm_init_callback; ncclAllGather of
pais_patch_result records on the context's stream) so that a C++ host bound as in INTEGRATION.md gets
multi-GPU with no Python.  This module only sets a job up from a Python launcher:

* `job_from_env()` reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torch.distributed.run, bench.py's own
  spawner) and opens a small gloo group for the control plane: handing out the RCCL unique id, barriers,
  the max-over-ranks of a timing.  The data path never goes through torch.
* `attach(m, job)` joins a driver to the job: RCCL when every rank has its own GPU; the callback transport
  (records through host memory, gloo all_gather) when several ranks share one GPU -- RCCL refuses that --
  which is how the 2-rank path is exercised on a 1-GPU box and on the CPU.
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Optional

import numpy as np

from .mvs import MVS, get_unique_id, UNIQUE_ID_BYTES


def shard_bounds(n: int, rank: int, world: int):
    """The contiguous, count-balanced shard of rank `rank` (the rule pais_mvs.hip applies)."""
    per = (n + world - 1) // world if n > 0 else 0
    lo = min(rank * per, n)
    hi = min(lo + per, n)
    return per, lo, hi


def shard_indices(n: int, rank: int, world: int):
    """The candidates of a batch of n that rank `rank` refines under PAIS_SHARD_STRIDED=1: candidate i goes to rank i mod world."""
    return list(range(rank, n, world))


@dataclass
class Job:
    rank: int
    world: int
    local_rank: int
    dist: object = None          # torch.distributed (gloo group) when world > 1

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()

    def max_over_ranks(self, v: float) -> float:
        if self.dist is None:
            return float(v)
        import torch
        t = torch.tensor([float(v)], dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def broadcast_bytes(self, b: Optional[bytes], n: int) -> bytes:
        if self.dist is None:
            return b
        import torch
        t = torch.zeros(n, dtype=torch.uint8)
        if self.rank == 0:
            t[:] = torch.frombuffer(bytearray(b), dtype=torch.uint8)
        self.dist.broadcast(t, src=0)
        return bytes(t.numpy().tobytes())

    def all_gather_bytes(self, send, nbytes: int) -> bytes:
        import torch
        t = torch.frombuffer(bytearray(bytes(send)), dtype=torch.uint8)
        outs = [torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(outs, t)
        return torch.cat(outs).numpy().tobytes()

    def close(self):
        if self.dist is not None:
            self.dist.destroy_process_group()
            self.dist = None


def job_from_env(force_group: bool = False) -> Job:
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    d = None
    if world > 1 or force_group:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if os.environ["MASTER_ADDR"] in ("127.0.0.1", "localhost"):
            os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")   # one node: the container's hostname may not resolve
        dist.init_process_group("gloo", rank=rank, world_size=world)   # control plane only
        d = dist
    return Job(rank, world, local, d)


def attach(m: MVS, job: Job, transport: str = "rccl"):
    """Join driver `m` to the job.  transport "rccl": ncclCommInitRank under the C ABI (one GPU per rank);
    "host": the callback transport, records through host memory with a gloo all_gather."""
    if job.world <= 1 and transport != "rccl":
        return
    if transport == "rccl":
        uid = get_unique_id() if job.rank == 0 else None
        uid = job.broadcast_bytes(uid, UNIQUE_ID_BYTES)
        m.comm_init_rccl(job.rank, job.world, uid)
    elif transport == "host":
        m.comm_init_callback(job.rank, job.world, job.all_gather_bytes)
    else:
        raise ValueError(transport)


def reconstruct(m: MVS, parents_per_round: int, max_rounds: int = 0):
    """MVS::refineSeedPatches + MVS::expansionPatches; sharded under the C ABI when a communicator is attached."""
    m.refineSeedPatches()
    m.expansionPatches(parents_per_round, max_rounds)
