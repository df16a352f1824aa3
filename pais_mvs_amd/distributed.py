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
    nccl = dist.get_backend() == "nccl"
    # buffers reused across rounds (grown on demand): pinned host staging for the candidates and the gathered records,
    # device buffers for this rank's shard and for the gathered result
    bufs = {"h_c": None, "d_c": None, "d_o": None, "d_all": None, "h_all": None}

    def ensure(name, nbytes, device, pinned=False):
        t = bufs[name]
        if t is None or t.numel() < nbytes:
            cap = int(nbytes * 1.5) + 4096
            t = torch.empty(cap, dtype=torch.uint8, device=device, pin_memory=pinned)
            bufs[name] = t
        return t

    def refine_into(d_o, cb: np.ndarray, count: int, has_seeds: bool, max_cam: int):
        """candidates (host bytes) -> records in d_o[:count*SZ_R] on the device"""
        if count == 0:
            return
        nb = count * SZ_C
        h_c = ensure("h_c", nb, "cpu", pinned=True)
        h_c[:nb].numpy()[:] = cb
        d_c = ensure("d_c", nb, dev)
        d_c[:nb].copy_(h_c[:nb], non_blocking=True)
        torch.cuda.current_stream().synchronize()
        _lib.check(L.pais_refine_batch_device(ctx, count, d_c.data_ptr(), d_o.data_ptr(), max_cam, 1 if has_seeds else 0),
                   "pais_refine_batch_device")
        _lib.check(L.pais_ctx_synchronize(ctx), "pais_ctx_synchronize")

    def to_host(d_t, nbytes):
        h = ensure("h_all", nbytes, "cpu", pinned=True)
        h[:nbytes].copy_(d_t[:nbytes], non_blocking=True)
        torch.cuda.current_stream().synchronize()
        return h[:nbytes].numpy()

    def run(cands_ptr, n, has_seeds, max_cam):
        cb = _cand_bytes(cands_ptr, n)
        if replicate_round(n, world):
            d_o = ensure("d_o", n * SZ_R, dev)
            refine_into(d_o, cb, n, has_seeds, max_cam)
            return to_host(d_o, n * SZ_R)
        per, lo, hi = shard_bounds(n, rank, world)
        d_o = ensure("d_o", per * SZ_R, dev)            # the tail of the last rank's shard is never read
        refine_into(d_o, cb[lo * SZ_C:hi * SZ_C], hi - lo, has_seeds, max_cam)
        if not nccl:                                    # test hook: gloo has no device collectives
            hb = d_o[:per * SZ_R].cpu()
            ho = torch.empty(world * per * SZ_R, dtype=torch.uint8)
            dist.all_gather_into_tensor(ho, hb)
            return ho.numpy()
        d_all = ensure("d_all", world * per * SZ_R, dev)
        dist.all_gather_into_tensor(d_all[:world * per * SZ_R], d_o[:per * SZ_R])   # the one collective of the round
        return to_host(d_all, world * per * SZ_R)

    ex = Exchange(rank, world, None, None)
    ex.run = run  # device-resident variant
    return ex
