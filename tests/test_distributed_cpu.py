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


def _worker(rank, world, port, q, replicate_below, fail_rank=-1, inject=None, shard_enum=False):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ["RANK"] = str(rank); os.environ["WORLD_SIZE"] = str(world)
    if shard_enum:   # (read when the driver is created: the listing of every round with a unit is dealt to the ranks)
        os.environ["PAIS_SHARD_ENUM"] = "1"; os.environ["PAIS_SHARD_ENUM_ABOVE"] = "0"
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
        if rank == fail_rank and not has_seeds:
            raise RuntimeError("injected failure of this rank's refinement")
        S.ptr.contents.cfg.neighborRadius = m.neighbor_radius()
        recs = _oracle_records(S, cands, n, has_seeds)
        C.memmove(out, recs, n * C.sizeof(_lib.PatchResult))
        calls["n"] += 1; calls["cands"] += n

    m.set_record_source(source)
    D.attach(m, job, transport="host")
    m.set_replicate_below(replicate_below)
    if inject and rank in inject[0]:
        m.test_inject(inject[1], inject[2])      # (include/pais_test_hooks.h: ring-retry status / failing growth / failing refinement)
    try:
        D.reconstruct(m, 8, max_rounds=10)
    except RuntimeError as e:
        q.put((rank, "error", str(e)))
        job.close()
        return
    cloud = m.cloud()
    st = m.stats()
    out = (rank, cloud.tobytes(), cloud.shape, int(st.candidates_effective), int(st.batches_sharded), int(st.batches_replicated),
           calls["cands"], int(st.exchange_bytes), [l.n for l in m.round_log()])
    q.put(out + ((int(st.rounds_enum_sharded), int(st.rounds)),) if shard_enum else out)
    job.close()


def _run(world, replicate_below, fail_rank=-1, inject=None, shard_enum=False):
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    port = _free_port()
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q, replicate_below, fail_rank, inject, shard_enum)) for r in range(world)]
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
    for rank, blob, shape, eff, sharded, replicated, cands, xbytes, nmin in two:
        assert shape == ref[2] and blob == ref[1] and eff == ref[3], rank
        assert sharded > 0 and replicated > 0, (sharded, replicated)
        # what travelled: wire slots (include/pais_hip.h), a fraction of the 1488-byte records
        assert 0 < xbytes < sharded * 2 * 64 + 0.3 * 1488 * (two[0][6] + two[1][6]), xbytes
    # the shards really were disjoint: together the two ranks refined fewer candidates than two full replicas would
    assert two[0][6] + two[1][6] < 2 * ref[6]
    # always shard (0): ragged shards incl. a last rank with padding; same cloud
    for rank, blob, shape, eff, sharded, replicated, cands, xbytes, nmin in _run(2, 0):
        assert blob == ref[1] and replicated == 0 and sharded > 0


@pytest.mark.parametrize("world", [4, 8])
def test_many_ranks_with_ragged_and_empty_shards(world):
    """4 and 8 ranks, always sharded: batches whose size is not a multiple of the world (a padded last shard) and batches
    with fewer candidates than ranks (ranks with NO candidate still take part in the exchange) -- same cloud as one rank."""
    ref = _run(1, 1024)[0]
    got = _run(world, 0)
    assert len(got) == world
    for rank, blob, shape, eff, sharded, replicated, cands, xbytes, nmin in got:
        assert shape == ref[2] and blob == ref[1] and eff == ref[3], rank
        assert replicated == 0 and sharded > 0
    ns = got[0][8]
    per = [(n + world - 1) // world for n in ns]
    assert any(n % world for n in ns)                          # ragged: a padded last shard
    if world == 8:
        assert any(p * (world - 1) >= n for n, p in zip(ns, per)), ns   # some batch left a rank without a candidate
    assert sum(g[6] for g in got) == ref[6]                   # disjoint shards: together exactly one replica's work


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_enumeration_gives_the_single_rank_cloud(world):
    """Round 6 (opt-in PAIS_SHARD_ENUM=1): the skip test + claim of a round's units dealt to the ranks by
    (camera, tile), the unit states merged with one all-gather per round, every rank building the same candidate list --
    same cloud, same candidates per round as one rank; sharded and replicated batches both occur behind it."""
    ref = _run(1, 1024)[0]
    got = _run(world, 72, shard_enum=True)
    assert len(got) == world
    for g in got:
        rank, blob, shape, eff, sharded, replicated, cands, xbytes, ns, (enum_rounds, rounds) = g
        assert shape == ref[2] and blob == ref[1] and eff == ref[3], rank
        assert ns == ref[8], rank                      # the same candidates in every round
        assert enum_rounds > 0 and enum_rounds <= rounds and sharded > 0, (enum_rounds, rounds, sharded)


def test_a_failing_rank_fails_every_rank_together():
    """a rank whose refinement fails must not leave the others waiting in the all-gather -- its status travels
    in the exchange header and every rank returns the error."""
    got = _run(2, 0, fail_rank=1)
    assert all(g[1] == "error" for g in got), got
    assert "rank 1" in got[0][2] or "record source" in got[0][2], got[0][2]


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_batch_protocol_second_exchange_growth_and_failure(world):
    """Round 6: the submit / finish protocol of a sharded batch -- status headers, growth handshake, the
    second exchange after a rank's k_pso_ring pass did not complete -- is ONE implementation (shard_submit / shard_finish,
    pais_mvs.hip) run over device memory + ncclAllGather or, here, over host memory + the caller's all-gather, world 2 and 4:
      * ranks whose header says "ring retry" refine their shard again and EVERY rank takes the second exchange: same cloud;
      * a rank whose exchange buffers cannot grow fails the growth handshake for every rank;
      * a rank whose refinement fails says so in its header: every rank returns the error, none waits in the collective."""
    ref = _run(1, 1024)[0]
    last = world - 1
    got = _run(world, 0, inject=((0, last), 1, 3))     # three batches each on the first and the last rank
    for rank, blob, shape, eff, sharded, replicated, cands, xbytes, nmin in got:
        assert shape == ref[2] and blob == ref[1] and eff == ref[3], rank
    # the retrying ranks refined those shards twice; every rank moved the bytes of the second exchanges
    assert got[0][6] > _run(world, 0)[0][6]
    got = _run(world, 0, inject=((last,), 2, 2))       # the second growth on the last rank
    assert all(g[1] == "error" for g in got), got
    assert all("rank %d could not grow" % last in g[2] or "injected" in g[2] for g in got), got
    got = _run(world, 0, inject=((0,), 3, 4))          # the fourth sharded refinement on rank 0
    assert all(g[1] == "error" for g in got), got
    assert all("rank 0" in g[2] or "injected" in g[2] for g in got), got


def test_record_wire_format_roundtrip():
    """pack -> unpack reproduces every byte of records whose arrays are empty beyond the batch's camera count; a slot is
    208 + 20 x roundup2(K) bytes."""
    from pais_mvs_amd import _lib
    L = _lib.load()
    L.pais_record_wire_bytes.restype = C.c_size_t
    L.pais_record_wire_bytes.argtypes = [C.c_int]
    L.pais_pack_records.argtypes = [C.c_int, C.POINTER(_lib.PatchResult), C.c_int, C.c_void_p]
    L.pais_unpack_records.argtypes = [C.c_int, C.c_void_p, C.c_int, C.POINTER(_lib.PatchResult)]
    assert L.pais_record_wire_bytes(5) == 208 + 20 * 6 and L.pais_record_wire_bytes(64) == 208 + 20 * 64 == 1488
    rng = np.random.default_rng(1)
    for K in (1, 5, 6, 31, 64):
        n = 7
        recs = (_lib.PatchResult * n)()
        raw = np.frombuffer(recs, dtype=np.uint8).reshape(n, C.sizeof(_lib.PatchResult))
        raw[:] = rng.integers(0, 256, raw.shape, dtype=np.uint8)
        for r in recs:                                           # arrays empty beyond K, as the batch calls leave them
            for i in range(K, 64):
                r.cam_idx[i] = 0; r.imgPoint[i][0] = 0.0; r.imgPoint[i][1] = 0.0
        want = bytes(raw.tobytes())
        wb = L.pais_record_wire_bytes(K)
        wire = (C.c_uint8 * (wb * n))()
        assert L.pais_pack_records(n, recs, K, wire) == 0
        back = (_lib.PatchResult * n)()
        assert L.pais_unpack_records(n, wire, K, back) == 0
        assert bytes(back) == want, K


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


def test_strided_shards_cover_everything_evenly():
    from pais_mvs_amd.distributed import shard_indices
    for n in (0, 1, 7, 8, 9, 1000):
        for world in (1, 2, 3, 8):
            parts = [shard_indices(n, r, world) for r in range(world)]
            assert sorted(i for p in parts for i in p) == list(range(n))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
            assert all(len(p) <= (n + world - 1) // world for p in parts)      # a rank's block never exceeds the wire slot


def test_strided_shards_rebuild_the_same_cloud(monkeypatch):
    """PAIS_SHARD_STRIDED=1 deals the same batch differently (candidate i to rank i mod world) and rebuilds the same cloud."""
    ref = _run(1, 1024)[0]
    monkeypatch.setenv("PAIS_SHARD_STRIDED", "1")      # (the spawned ranks inherit the environment)
    for rank, blob, shape, eff, sharded, replicated, cands, xbytes, nmin in _run(2, 0):
        assert blob == ref[1] and shape == ref[2] and sharded > 0


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
