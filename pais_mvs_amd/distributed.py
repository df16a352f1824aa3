"""Multi-GPU reconstruction: one process per GPU, replicated scheduler, sharded refinement.

Every rank holds the full scene in its own HBM and an identical driver (pais_mvs).
Per round (and once for the seeds) each rank refines a contiguous shard of the
round's candidate list on its GPU; the fixed-size result records are exchanged with
ONE all-gather (RCCL over xGMI when the backend is "nccl"); every rank then runs the
same deterministic host replay, so cell maps / queue stay replicated without further
traffic.  The accepted cloud is bit-identical for any world size.
"""
from __future__ import annotations

import ctypes as C
from typing import Callable, Optional

import numpy as np

from . import _lib
from .mvs import MVS

SZ_C = C.sizeof(_lib.Candidate)
SZ_R = C.sizeof(_lib.PatchResult)


def shard_bounds(n: int, rank: int, world: int):
    per = (n + world - 1) // world if n > 0 else 0
    lo = min(rank * per, n)
    hi = min(lo + per, n)
    return per, lo, hi


REPLICATE_BELOW_PER_RANK = 64


def replicate_round(n: int, world: int) -> bool:
    """Rounds with fewer than 64 candidates per rank are latency bound on one GPU already: sharding them cannot
    make them faster and the all-gather + synchronisation only adds to the round.  Every rank then refines the whole
    round itself -- the refinement is deterministic, so the replicas agree bit for bit and no exchange is needed."""
    return world > 1 and n < REPLICATE_BELOW_PER_RANK * world


def _cand_bytes(cands_ptr, n: int) -> np.ndarray:
    if n == 0:
        return np.zeros(0, dtype=np.uint8)
    return np.ctypeslib.as_array(C.cast(cands_ptr, C.POINTER(C.c_uint8)), shape=(n * SZ_C,))


class Exchange:
    """refine_shard(cand_bytes[lo:hi], count, has_seeds, max_cam) -> uint8 array of count records;
    all_gather(padded uint8 array of per*SZ_R) -> uint8 array of world*per*SZ_R."""

    def __init__(self, rank: int, world: int, refine_shard: Callable, all_gather: Callable):
        self.rank, self.world = rank, world
        self.refine_shard, self.all_gather = refine_shard, all_gather

    def run(self, cands_ptr, n: int, has_seeds: bool, max_cam: int):
        cb = _cand_bytes(cands_ptr, n)
        if replicate_round(n, self.world):
            # thin round: every rank refines all of it (bit-identical results on every GPU), no collective
            return np.frombuffer(self.refine_shard(cb, n, has_seeds, max_cam), dtype=np.uint8, count=n * SZ_R).copy()
        per, lo, hi = shard_bounds(n, self.rank, self.world)
        mine = self.refine_shard(cb[lo * SZ_C:hi * SZ_C], hi - lo, has_seeds, max_cam)
        buf = np.zeros(per * SZ_R, dtype=np.uint8)
        if hi > lo:
            buf[:(hi - lo) * SZ_R] = np.frombuffer(mine, dtype=np.uint8, count=(hi - lo) * SZ_R)
        allb = np.ascontiguousarray(self.all_gather(buf))
        return allb   # first n records (shards are contiguous, in rank order) are the round's results


def reconstruct(m: MVS, parents_per_round: int, ex: Exchange, max_rounds: int = 0):
    """MVS::refineSeedPatches + MVS::expansionPatches with sharded refinement."""
    ncam = len(m.cameras)
    cands, n = m.seed_begin()
    if n:
        allb = ex.run(cands, n, True, min(ncam, _lib.MAX_VIS))
        m.seed_commit(C.cast(allb.ctypes.data, C.POINTER(_lib.PatchResult)), n)
    m.expansion_begin()
    rounds = 0
    while True:
        done, cands, n = m.round_begin(parents_per_round)
        if done:
            break
        if n:
            allb = ex.run(cands, n, False, min(ncam, _lib.MAX_VIS))
            m.round_commit(C.cast(allb.ctypes.data, C.POINTER(_lib.PatchResult)), n)
        else:
            m.round_commit(None, 0)
        rounds += 1
        if max_rounds and rounds >= max_rounds:
            break
    m.expansion_end()


def torch_gpu_exchange(m: MVS, rank: int, world: int) -> Exchange:
    """Refinement on this rank's GPU through the C ABI's device entry point; all-gather
    through torch.distributed (backend "nccl" == RCCL) on device buffers."""
    import torch
    import torch.distributed as dist

    L = m.L
    ctx = m.ctx_handle
    dev = torch.device("cuda", torch.cuda.current_device())

    def refine_shard(cb: np.ndarray, count: int, has_seeds: bool, max_cam: int):
        if count == 0:
            return None
        d_c = torch.from_numpy(np.ascontiguousarray(cb)).to(dev)
        d_o = torch.empty(count * SZ_R, dtype=torch.uint8, device=dev)
        torch.cuda.current_stream().synchronize()
        _lib.check(L.pais_refine_batch_device(ctx, count, d_c.data_ptr(), d_o.data_ptr(), max_cam, 1 if has_seeds else 0),
                   "pais_refine_batch_device")
        _lib.check(L.pais_ctx_synchronize(ctx), "pais_ctx_synchronize")
        return d_o

    def run(cands_ptr, n, has_seeds, max_cam):
        cb = _cand_bytes(cands_ptr, n)
        if replicate_round(n, world):
            return refine_shard(cb, n, has_seeds, max_cam).cpu().numpy()
        per, lo, hi = shard_bounds(n, rank, world)
        d_o = refine_shard(cb[lo * SZ_C:hi * SZ_C], hi - lo, has_seeds, max_cam)
        buf = torch.zeros(per * SZ_R, dtype=torch.uint8, device=dev)
        if d_o is not None:
            buf[:(hi - lo) * SZ_R] = d_o
        if dist.get_backend() != "nccl":           # test hook: gloo has no device collectives
            hb = buf.cpu()
            ho = torch.empty(world * per * SZ_R, dtype=torch.uint8)
            dist.all_gather_into_tensor(ho, hb)
            return ho.numpy()
        out = torch.empty(world * per * SZ_R, dtype=torch.uint8, device=dev)
        dist.all_gather_into_tensor(out, buf)      # the one collective of the round
        return out.cpu().numpy()

    ex = Exchange(rank, world, None, None)
    ex.run = run  # device-resident variant
    return ex
