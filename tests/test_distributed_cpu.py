"""The N>1 path on the CPU: 2 ranks over gloo.  Each rank owns a replicated scheduler
(device = -1), refines its contiguous shard of every round with the ORACLE as a stand-in for its
GPU, exchanges the records with one all_gather per round, and replays.  Both ranks must end with
the cloud a single rank produces."""
import ctypes as C
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pais_mvs_amd import synth, _lib
    from pais_mvs_amd.config import readme_config
    from pais_mvs_amd.mvs import MVS
    from pais_mvs_amd import distributed as D
    from tests import common
    from tests.test_scheduler_cpu import _oracle_records
    scene = synth.pawn_scene(width=320, height=240, n_seeds=24)
    cfg = readme_config(particleNum=6, maxIteration=8)
    S = common.oracle_scene(cfg, scene)
    m = MVS(cfg, scene.cameras, device=-1, seed=42)
    for X, vis in scene.seeds:
        m.add_seed(X, vis)

    def refine_shard(cb, count, has_seeds, max_cam):
        if count == 0:
            return b""
        S.ptr.contents.cfg.neighborRadius = m.neighbor_radius()
        arr = (_lib.Candidate * count).from_buffer_copy(cb.tobytes())
        recs = _oracle_records(S, arr, count, has_seeds)
        return bytes(recs)[:count * D.SZ_R]

    def all_gather(buf):
        t = torch.from_numpy(buf.copy())
        outs = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(outs, t)
        return torch.cat(outs).numpy()

    # the product replicates thin rounds (fewer than 64 candidates per rank) instead of sharding them; here the choice is
    # made by the parity of the round size so that both paths certainly run (any rule works as long as all ranks agree)
    stats = {"sharded": 0, "replicated": 0}

    def rr(n, w):
        r = w > 1 and n % 2 == 0
        stats["replicated" if r else "sharded"] += 1
        return r
    D.replicate_round = rr
    ex = D.Exchange(rank, world, refine_shard, all_gather)
    D.reconstruct(m, 8, ex, max_rounds=10)
    q.put(("stats", rank, dict(stats)))
    cloud = m.cloud()
    st = m.stats()
    q.put((rank, cloud.tobytes(), cloud.shape, int(st.candidates_effective)))
    dist.destroy_process_group()


def test_two_rank_sharded_reconstruction_is_rank_count_invariant():
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    res = {}
    for world in (1, 2):
        port = _free_port()
        q = ctx.Queue()
        ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
        for p in ps:
            p.start()
        got = [q.get(timeout=300) for _ in range(2 * world)]
        for p in ps:
            p.join(timeout=60)
            assert p.exitcode == 0
        res[world] = [g for g in got if g[0] != "stats"]
        if world == 2:   # both the sharded and the replicated kind of round were exercised on every rank
            for g in got:
                if g[0] == "stats":
                    assert g[2]["sharded"] > 0 and g[2]["replicated"] > 0, g
    ref = res[1][0]
    assert ref[2][0] > 24
    for rank, blob, shape, eff in res[2]:
        assert shape == ref[2] and blob == ref[1] and eff == ref[3]


def test_shard_bounds_cover_everything():
    from pais_mvs_amd.distributed import shard_bounds
    for n in (0, 1, 7, 8, 9, 1000):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                per, lo, hi = shard_bounds(n, r, world)
                assert hi - lo <= per
                seen += list(range(lo, hi))
            assert seen == list(range(n))
