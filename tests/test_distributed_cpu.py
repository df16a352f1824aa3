"""The N>1 path on the CPU, THROUGH THE C ENTRY POINTS: 2 ranks, each with a replicated GPU-less driver
(device = -1) whose records come from the ORACLE (pais_mvs_set_record_source -- the stand-in for the rank's GPU)
and whose per-batch all-gather is gloo (pais_mvs_comm_init_callback).  pais_mvs_refine_seed_patches /
pais_mvs_expansion_patches shard, exchange and replay inside libpais_hip.so; both ranks must end with the cloud a
single rank produces, byte for byte."""
import ctypes as C
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q, replicate_below):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ["RANK"] = str(rank); os.environ["WORLD_SIZE"] = str(world)
    from pais_mvs_amd import synth, _lib
    from pais_mvs_amd.config import readme_config
    from pais_mvs_amd.mvs import MVS
    from pais_mvs_amd import distributed as D
    from tests import common
    from tests.test_scheduler_cpu import _oracle_records
    job = D.job_from_env(force_group=True)
    scene = synth.pawn_scene(width=320, height=240, n_seeds=24)
    cfg = readme_config(particleNum=6, maxIteration=8)
    S = common.oracle_scene(cfg, scene)
    m = MVS(cfg, scene.cameras, device=-1, seed=42)
    for X, vis in scene.seeds:
        m.add_seed(X, vis)
    calls = {"n": 0, "cands": 0}

    def source(n, cands, out, has_seeds):
        S.ptr.contents.cfg.neighborRadius = m.neighbor_radius()
        recs = _oracle_records(S, cands, n, has_seeds)
        C.memmove(out, recs, n * C.sizeof(_lib.PatchResult))
        calls["n"] += 1; calls["cands"] += n

    m.set_record_source(source)
    D.attach(m, job, transport="host")
    m.set_replicate_below(replicate_below)
    D.reconstruct(m, 8, max_rounds=10)
    cloud = m.cloud()
    st = m.stats()
    q.put((rank, cloud.tobytes(), cloud.shape, int(st.candidates_effective), int(st.batches_sharded), int(st.batches_replicated),
           calls["cands"]))
    job.close()


def _run(world, replicate_below):
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    port = _free_port()
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q, replicate_below)) for r in range(world)]
    for p in ps:
        p.start()
    got = [q.get(timeout=600) for _ in range(world)]
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    return sorted(got)


def test_two_rank_sharded_reconstruction_is_rank_count_invariant():
    ref = _run(1, 1024)[0]
    assert ref[2][0] > 24 and ref[4] == 0 and ref[5] == 0         # one rank: nothing sharded, nothing replicated
    # replicate_below = 72 evaluation waves (6 particles: 12 expansion candidates): thinner batches are replicated, the
    # others sharded -- both kinds of batch must occur so that both code paths run
    two = _run(2, 72)
    for rank, blob, shape, eff, sharded, replicated, cands in two:
        assert shape == ref[2] and blob == ref[1] and eff == ref[3], rank
        assert sharded > 0 and replicated > 0, (sharded, replicated)
    # the shards really were disjoint: together the two ranks refined fewer candidates than two full replicas would
    assert two[0][6] + two[1][6] < 2 * ref[6]
    # always shard (0): ragged shards incl. a last rank with padding; same cloud
    for rank, blob, shape, eff, sharded, replicated, cands in _run(2, 0):
        assert blob == ref[1] and replicated == 0 and sharded > 0


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


def test_rank_arguments_are_validated(pawn_small):
    from pais_mvs_amd.config import readme_config
    from pais_mvs_amd.mvs import MVS
    m = MVS(readme_config(), pawn_small.cameras, device=-1)
    with pytest.raises(RuntimeError):
        m.comm_init_rccl(0, 2, b"\0" * 128)          # no GPU: RCCL cannot be attached
    with pytest.raises(RuntimeError):
        m.comm_init_callback(2, 2, lambda send, n: b"")   # rank out of range
    m.comm_init_callback(0, 1, lambda send, n: bytes(send))
    with pytest.raises(RuntimeError):
        m.comm_init_callback(0, 1, lambda send, n: bytes(send))   # already attached
    m.close()
