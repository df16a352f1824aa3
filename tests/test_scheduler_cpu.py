"""Host scheduler (pais_mvs.hip: cell maps, queue, speculative rounds + exact replay) on the CPU.

The driver is created without a GPU context (device = -1) and fed, through the same
stepwise entry points the multi-GPU path uses, with records computed by the ORACLE.
Its accepted cloud must equal the oracle's own round-based expansion R(B)
(po_mvs_expansion_patches), which for B = 1 is the reference loop mvs.cpp:233-275.
This pins candidate enumeration, skipNeighborCell / runtimeFiltering / insertPatch
replay, queue order and tie-breaks -- everything around the kernels.
"""
import ctypes as C

import numpy as np
import pytest

from tests import common


def _oracle_records(S, cands_ptr, n, is_seed):
    """Refine product candidates with the oracle and convert to pais_patch_result records."""
    from oracle import po
    from pais_mvs_amd import _lib
    L = po.lib()
    out = (_lib.PatchResult * max(n, 1))()
    for k in range(n):
        c = cands_ptr[k]
        p = po.Patch()
        # re-create the constructed patch state on the oracle side from the candidate
        p.id = -1; p.refCamIdx = -1; p.LOD = -1
        p.fitness = common.DBL_MAX; p.priority = common.DBL_MAX
        p.type = c.type
        p.center[:] = c.center[:]
        p.normal[:] = c.normal[:]
        p.normalS[:] = c.normalS[:]
        p.numCam = c.num_cam
        for i in range(c.num_cam):
            p.camIdx[i] = c.cam_idx[i]
        p.key = c.key
        if is_seed:
            L.po_refine_seed(S.ptr, C.byref(p))
        else:
            if p.numCam < S.cfg.minCamNum:
                p.drop = 1          # expandVisibleCamera :758-760
            L.po_refine(S.ptr, C.byref(p))
            L.po_remove_invisible_camera(S.ptr, C.byref(p))
        r = out[k]
        r.center[:] = p.center[:]; r.normal[:] = p.normal[:]; r.normalS[:] = p.normalS[:]
        r.ray[:] = p.ray[:]; r.depth = p.depth; r.depthRange[:] = p.depthRange[:]
        r.fitness = p.fitness; r.priority = p.priority; r.correlation = p.correlation
        for i in range(p.numCam):
            r.imgPoint[i][0] = p.imgPoint[i][0]; r.imgPoint[i][1] = p.imgPoint[i][1]
            r.cam_idx[i] = p.camIdx[i]
        r.key = p.key; r.type = p.type; r.dropped = p.drop; r.num_cam = p.numCam
        r.ref_cam = p.refCamIdx; r.lod = p.LOD
        r.pso_runs = p.psoRuns; r.pso_iterations = p.psoIters; r.pso_evals = p.psoEvals
    return out


def _run_product_with_oracle_records(cfg, scene, B, max_rounds, thin=None):
    from pais_mvs_amd.mvs import MVS
    S = common.oracle_scene(cfg, scene)
    m = MVS(cfg, scene.cameras, device=-1, seed=42)
    if thin is not None:
        m.set_thin_front(thin)
    for X, vis in scene.seeds:
        m.add_seed(X, vis)
    cands, n = m.seed_begin()
    S.ptr.contents.cfg.neighborRadius = m.neighbor_radius()
    m.seed_commit(_oracle_records(S, cands, n, True), n)
    m.expansion_begin()
    S.ptr.contents.cfg.neighborRadius = m.neighbor_radius()
    rounds = 0
    while True:
        done, cands, n = m.round_begin(B)
        if done:
            break
        m.round_commit(_oracle_records(S, cands, n, False), n)
        rounds += 1
        if max_rounds and rounds >= max_rounds:
            break
    m.expansion_end()
    return m


def _run_oracle(cfg, scene, B, max_rounds, thin=None):
    from oracle import po
    S = common.oracle_scene(cfg, scene)
    L = po.lib()
    mo = L.po_mvs_create(S.ptr)
    for X, vis in scene.seeds:
        L.po_mvs_add_seed(mo, po.darr(X), len(vis), po.iarr(vis))
    L.po_mvs_refine_seed_patches(mo)
    if thin is not None:
        L.po_mvs_set_thin_front(mo, thin)
    L.po_mvs_expansion_patches(mo, B, max_rounds, 1)
    pats = []
    for i in range(L.po_mvs_num_slots(mo)):
        pp = L.po_mvs_get_patch(mo, i)
        if pp:
            p = pp.contents
            pats.append((list(p.center[:]), list(p.normal[:]), p.cams(), p.fitness, p.priority))
    calls = L.po_mvs_refine_calls(mo)
    L.po_mvs_destroy(mo)
    return pats, calls, S


# thin: the thin-front threshold of the round rule (None = the shared default, 0 = one camera slot per round
# always, 3 = switches between both kinds of round with B = 8, 10**6 = whole parents always)
@pytest.mark.parametrize("B,max_rounds,thin", [(1, 60, None), (8, 25, None), (1, 60, 0), (8, 25, 0), (8, 25, 3), (8, 12, 10 ** 6)])
def test_scheduler_reproduces_oracle_rounds(pawn_small, B, max_rounds, thin):
    from pais_mvs_amd.config import readme_config
    cfg = readme_config(particleNum=6, maxIteration=8)   # small swarm: this test is about the scheduler
    want, oracle_calls, _S = _run_oracle(cfg, pawn_small, B, max_rounds, thin)
    m = _run_product_with_oracle_records(cfg, pawn_small, B, max_rounds, thin)
    got = [(list(p.center[:]), list(p.normal[:]), p.cams(), p.fitness, p.priority) for p in m.patches()]
    assert len(got) == len(want) and len(got) > len(pawn_small.seeds) // 2
    for a, b in zip(got, want):
        assert a == b
    st = m.stats()
    # the sequential order evaluates exactly the effective candidates (+ seeds)
    assert st.candidates_effective + st.seeds_refined == oracle_calls
    assert st.candidates_refined >= st.candidates_effective
    m.close()


@pytest.mark.parametrize("strategy", [1, 2, 3])
@pytest.mark.parametrize("B,max_rounds", [(1, 80), (8, 20)])
def test_expansion_strategies_reproduce_oracle(pawn_small, strategy, B, max_rounds):
    """MvsConfig::expansionStrategy 1 worst-first (mvs.cpp:695-732), 2 breadth-first (:734-759), 3 depth-first (:761-788,
    which never examines queue[0]): the driver's containers (heap with lazy deletion / deque) against the oracle's literal
    vector<int> scans, patch by patch, for the reference's one-parent order and for rounds of 8."""
    from pais_mvs_amd.config import readme_config
    cfg = readme_config(particleNum=6, maxIteration=8, expansionStrategy=strategy)
    want, oracle_calls, _S = _run_oracle(cfg, pawn_small, B, max_rounds)
    m = _run_product_with_oracle_records(cfg, pawn_small, B, max_rounds)
    got = [(list(p.center[:]), list(p.normal[:]), p.cams(), p.fitness, p.priority) for p in m.patches()]
    assert len(got) == len(want) and len(got) > len(pawn_small.seeds) // 2
    for a, b in zip(got, want):
        assert a == b
    st = m.stats()
    assert st.candidates_effective + st.seeds_refined == oracle_calls
    m.close()


def test_strategies_pop_in_their_own_order(pawn_small):
    """The four strategies really are different schedules: after the same number of one-parent rounds the clouds differ,
    and the popped parents follow the strategy (best-first pops ascending priority among the seeds, worst-first descending,
    breadth-first in insertion order, depth-first from the back without ever taking the first seed while others remain)."""
    from oracle import po
    from pais_mvs_amd.config import readme_config
    L = po.lib()
    clouds = {}
    for strategy in (0, 1, 2, 3):
        cfg = readme_config(particleNum=6, maxIteration=8, expansionStrategy=strategy)
        pats, _, _ = _run_oracle(cfg, pawn_small, 1, 6)
        clouds[strategy] = pats
    assert len({str(v) for v in clouds.values()}) == 4


def test_queue_tail_quirks_of_the_reference_loop(pawn_small):
    """mvs.cpp:241-272 pops BEFORE testing `while (!queue.empty())`, so the parent whose pop empties the queue is never
    expanded; depth-first (mvs.cpp:761-788) scans from the back and stops at queue.begin() without examining it, so the
    FIRST queued patch is never expanded while any other is live.  Children are suppressed here (every candidate record
    is `dropped`), which leaves exactly the seeds in the queue."""
    from pais_mvs_amd import _lib
    from pais_mvs_amd.config import readme_config
    from pais_mvs_amd.mvs import MVS
    for strategy in (0, 1, 2, 3):
        cfg = readme_config(particleNum=6, maxIteration=8, expansionStrategy=strategy)
        S = common.oracle_scene(cfg, pawn_small)
        m = MVS(cfg, pawn_small.cameras, device=-1, seed=42)
        for X, vis in pawn_small.seeds:
            m.add_seed(X, vis)
        cands, n = m.seed_begin()
        S.ptr.contents.cfg.neighborRadius = m.neighbor_radius()
        m.seed_commit(_oracle_records(S, cands, n, True), n)
        m.expansion_begin()
        while True:
            done, cands, n = m.round_begin(1)
            if done:
                break
            recs = (_lib.PatchResult * max(n, 1))()
            for k in range(n):
                recs[k].dropped = 1
            m.round_commit(recs, n)
        alive, unexpanded = [], []
        for i in range(m.num_slots()):
            r = _lib.PatchResult(); e = C.c_int(0)
            if m.L.pais_mvs_get_patch(m.h, i, C.byref(r), C.byref(e)) == 0:
                alive.append((i, r.priority))
                if not e.value:
                    unexpanded.append(i)
        assert len(alive) >= 10
        # parents whose runtimeFiltering failed at pop time were deleted; of the rest exactly one stays unexpanded
        assert len(unexpanded) == 1, (strategy, unexpanded)
        ids = [i for i, _ in alive]
        if strategy == 0:      # best-first: the worst priority is popped last
            assert unexpanded[0] == max(alive, key=lambda t: (t[1], t[0]))[0]
        elif strategy == 1:    # worst-first: the best priority is popped last
            assert unexpanded[0] == min(alive, key=lambda t: (t[1], -t[0]))[0]
        elif strategy == 2:    # breadth-first: the last queued
            assert unexpanded[0] == ids[-1]
        else:                  # depth-first: queue[0], never examined
            assert unexpanded[0] == ids[0]
        m.close()


def test_one_parent_per_round_is_the_reference_order_for_any_thin_front(pawn_small):
    """B = 1: taking a parent's camera slots one round at a time or all in one round is the same sequential
    loop (mvs.cpp:529-563), so the cloud must not depend on the thin-front threshold."""
    from pais_mvs_amd.config import readme_config
    cfg = readme_config(particleNum=6, maxIteration=8)
    a, calls_a, _ = _run_oracle(cfg, pawn_small, 1, 0, 0)
    b, calls_b, _ = _run_oracle(cfg, pawn_small, 1, 0, 64)
    assert a == b and calls_a == calls_b and len(a) > len(pawn_small.seeds)


def test_recentering_of_nvm_seeds(pawn_small):
    """N4: MVS::reCentering (patch.cpp:67-112) -- the product's host statement equals the oracle's bit for bit, and with
    exact measurements it returns the point the measurements came from."""
    from oracle import po
    from pais_mvs_amd.config import readme_config
    from pais_mvs_amd.mvs import MVS
    cfg = readme_config()
    S = common.oracle_scene(cfg, pawn_small)
    L = po.lib()
    m = MVS(cfg, pawn_small.cameras, device=-1, seed=42)
    rng = np.random.default_rng(11)
    for X, vis in pawn_small.seeds[:40]:
        X = np.asarray(X, float)
        pts = []
        for c in vis:
            cam = pawn_small.cameras[c]
            q = cam.rotation @ X + cam.translation
            pts.append([cam.focal[0] * q[0] / q[2] + cam.principle_point[0], cam.focal[1] * q[1] / q[2] + cam.principle_point[1]])
        start = X + rng.normal(0, 0.05, 3)                    # what an SfM point looks like before re-triangulation
        for noise in (0.0, 0.7):
            meas = [[p[0] + rng.normal(0, noise), p[1] + rng.normal(0, noise)] for p in pts]
            flat = [v for p in meas for v in p]
            want = (C.c_double * 3)()
            L.po_recenter(S.ptr, len(vis), po.iarr(vis), po.darr(flat), want)
            pid = m.add_seed_measured(start, vis, meas, recenter=True)
            got = m.get_patch(pid)
            assert list(got.center[:]) == list(want[:])
            assert [list(got.imgPoint[k]) for k in range(len(vis))] == meas
            if noise == 0.0:
                assert np.linalg.norm(np.array(got.center[:]) - X) < 1e-9
        pid = m.add_seed_measured(start, vis, pts, recenter=False)
        assert list(m.get_patch(pid).center[:]) == list(start)
    m.close()
