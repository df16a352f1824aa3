"""Post filters (the `-f` verb, TMVS.cpp:124-172; MVS::cellFiltering / visibilityFiltering / neighborCellFiltering /
neighborPatchFiltering, mvs.cpp:278-524) -- SURVEY 8(f) N3, the step after the hot path.

A cloud is grown by the ORACLE's expansion, written down the way a .mvs file holds it (centre, spherical normal,
cameras, fitness, correlation) and loaded through the loader constructor (patch.cpp:45-59) into a fresh oracle
driver and into the product's driver.  The host filters of the product must then delete exactly the patches the
oracle's restatement deletes, pass by pass; the all-pairs neighbour counts of the PCMVS filter are a HIP kernel
(GPU test) and have no host path.
"""
import ctypes as C

import numpy as np
import pytest

from tests import common


def _grow_cloud(cfg, scene, max_rounds=40):
    from oracle import po
    S = common.oracle_scene(cfg, scene)
    L = po.lib()
    mo = L.po_mvs_create(S.ptr)
    for X, vis in scene.seeds:
        L.po_mvs_add_seed(mo, po.darr(X), len(vis), po.iarr(vis))
    L.po_mvs_refine_seed_patches(mo)
    L.po_mvs_expansion_patches(mo, 64, max_rounds, 1)
    cloud = []
    for i in range(L.po_mvs_num_slots(mo)):
        pp = L.po_mvs_get_patch(mo, i)
        if pp:
            p = pp.contents
            cloud.append((list(p.center[:]), list(p.normalS[:]), p.cams(), p.fitness, p.correlation))
    L.po_mvs_destroy(mo)
    return cloud


def _load_oracle(cfg, scene, cloud):
    from oracle import po
    S = common.oracle_scene(cfg, scene)
    L = po.lib()
    mo = L.po_mvs_create(S.ptr)
    for c, ns, cams, fit, corr in cloud:
        L.po_mvs_load_patch(mo, po.darr(c), po.darr(ns), len(cams), po.iarr(cams), fit, corr)
    return S, L, mo


def _alive_oracle(L, mo):
    return [i for i in range(L.po_mvs_num_slots(mo)) if L.po_mvs_get_patch(mo, i)]


def _alive_product(m):
    return [i for i in range(m.num_slots()) if m.get_patch(i) is not None]


@pytest.fixture(scope="module")
def cloud(pawn_small):
    from pais_mvs_amd.config import readme_config
    # small swarm (this is about the filters); a loose cell budget and correlation gate so that cells hold several
    # patches and the filters have something to remove
    cfg = readme_config(particleNum=6, maxIteration=8, maxCellPatchNum=6, minCorrelation=0.6)
    return cfg, _grow_cloud(cfg, pawn_small)


def test_host_filters_delete_what_the_oracle_deletes(pawn_small, cloud):
    from pais_mvs_amd.mvs import MVS
    cfg, cl = cloud
    assert len(cl) > 200
    S, L, mo = _load_oracle(cfg, pawn_small, cl)
    m = MVS(cfg, pawn_small.cameras, device=-1, seed=42)
    for c, ns, cams, fit, corr in cl:
        m.load_patch(c, ns, cams, fit, corr)
    # the loader constructor: same normals and image points on both sides
    for i in (0, len(cl) // 2, len(cl) - 1):
        po_p = L.po_mvs_get_patch(mo, i).contents
        pr = m.get_patch(i)
        assert list(pr.normal[:]) == list(po_p.normal[:])
        for k in range(po_p.numCam):
            assert pr.imgPoint[k][0] == po_p.imgPoint[k][0] and pr.imgPoint[k][1] == po_p.imgPoint[k][1]
    removed = []
    n0 = len(cl)
    L.po_mvs_cell_filtering(mo); m.cellFiltering()
    a = _alive_oracle(L, mo)
    assert a == _alive_product(m)
    removed.append(n0 - len(a)); n0 = len(a)
    L.po_mvs_visibility_filtering(mo); m.visibilityFiltering()
    a = _alive_oracle(L, mo)
    assert a == _alive_product(m)
    removed.append(n0 - len(a)); n0 = len(a)
    L.po_mvs_neighbor_cell_filtering(mo, 0.25); m.neighborCellFiltering(0.25)
    a = _alive_oracle(L, mo)
    assert a == _alive_product(m)
    removed.append(n0 - len(a))
    assert m.neighbor_radius() == S.ptr.contents.cfg.neighborRadius
    assert sum(removed) > 0 and len(a) > 0, removed          # the passes did something and left something
    # the PCMVS filter's pair counts are a HIP kernel: without a device it must fail loudly, not fall back
    with pytest.raises(RuntimeError):
        m.neighborPatchFiltering(0.25)
    m.close()
    L.po_mvs_destroy(mo)


@pytest.mark.gpu
def test_filter_chain_on_gpu_matches_oracle(pawn_small, cloud):
    from pais_mvs_amd.mvs import MVS
    cfg, cl = cloud
    S, L, mo = _load_oracle(cfg, pawn_small, cl)
    m = MVS(cfg, pawn_small.cameras, device=0, seed=42)
    for c, ns, cams, fit, corr in cl:
        m.load_patch(c, ns, cams, fit, corr)
    # all-pairs neighbour counts alone, against the oracle's O(n^2) loop (before any deletion)
    n = len(cl)
    cen = np.ascontiguousarray(np.array([c for c, *_ in cl], dtype=np.float64))
    counts = np.zeros(n, dtype=np.int32)
    L.po_mvs_set_neighbor_radius(mo)
    radius = S.ptr.contents.cfg.neighborRadius
    rc = m.L.pais_neighbor_count(m.ctx_handle, n, cen.ctypes.data_as(C.POINTER(C.c_double)), C.c_double(radius),
                                 counts.ctypes.data_as(C.POINTER(C.c_int32)), None)
    assert rc == 0
    want = np.array([int(np.sum(np.sqrt(((cen - cen[i]) ** 2)[:, 0] + ((cen - cen[i]) ** 2)[:, 1] + ((cen - cen[i]) ** 2)[:, 2]) <= radius)) - 1
                     for i in range(n)], dtype=np.int32)
    assert np.array_equal(counts, want)
    # the whole `-f` chain: PMVS filters then the PCMVS filter
    L.po_mvs_cell_filtering(mo); m.cellFiltering()
    L.po_mvs_visibility_filtering(mo); m.visibilityFiltering()
    L.po_mvs_neighbor_cell_filtering(mo, 0.25); m.neighborCellFiltering(0.25)
    before = len(_alive_product(m))
    L.po_mvs_neighbor_patch_filtering(mo, 0.25, None)
    ms = m.neighborPatchFiltering(0.25)
    a = _alive_oracle(L, mo)
    assert a == _alive_product(m) and 0 < len(a) <= before and ms >= 0.0
    m.close()
    L.po_mvs_destroy(mo)
