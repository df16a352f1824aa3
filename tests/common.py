"""Shared helpers of the parity tests: build oracle scene + product context from one synthetic scene."""
from __future__ import annotations

import ctypes as C
import math

import numpy as np

from oracle import po
from pais_mvs_amd import _lib
from pais_mvs_amd.config import MvsConfig, readme_config
from pais_mvs_amd.context import make_candidate

DBL_MAX = 1.7976931348623157e308


def oracle_cfg(cfg: MvsConfig) -> po.Config:
    c = po.Config()
    for name, _ in po.Config._fields_:
        if name == "patchSize":
            continue
        v = getattr(cfg, name)
        setattr(c, name, int(v) if isinstance(v, bool) else v)
    c.patchSize = cfg.patchSize
    return c


def oracle_scene(cfg: MvsConfig, scene, seed=42) -> po.OracleScene:
    return po.OracleScene(oracle_cfg(cfg), scene.cameras, seed=seed)


def rel_l2(a, b) -> float:
    a = np.asarray(a, float); b = np.asarray(b, float)
    d = np.linalg.norm(a - b)
    n = np.linalg.norm(b)
    return float(d / n) if n > 0 else float(d)


def seed_candidates(S: po.OracleScene, scene):
    """Seed constructor on the oracle side (patch.cpp:26-34) -> (oracle patches, product candidates)."""
    pats, cands = [], []
    for i, (X, vis) in enumerate(scene.seeds):
        p = S.seed_patch(X, vis, key=i)
        pats.append(p)
        cands.append(make_candidate(p.center[:], p.normal[:], p.cams(), i, 0,
                                    normalS=p.normalS[:]))
    return pats, cands


def same_value(a: float, b: float, rtol: float) -> bool:
    if math.isnan(a) or math.isnan(b):
        return math.isnan(a) and math.isnan(b)
    if a == b:
        return True
    return abs(a - b) <= rtol * max(abs(a), abs(b))


def oracle_reconstruct(cfg, scene, B, max_rounds=0, parallel=True, kernel_arithmetic=True, thin_front=None, seed=42):
    """Seeds + expansion through the oracle's drivers (po_mvs_refine_seed_patches / po_mvs_expansion_patches).
    Returns (rows for patches_sha1, refine calls, accepted patches, speculative records of the parallel mode)."""
    S = oracle_scene(cfg, scene, seed=seed)
    S.set_kernel_arithmetic(kernel_arithmetic)
    L = po.lib()
    mo = L.po_mvs_create(S.ptr)
    L.po_mvs_set_parallel(mo, 1 if parallel else 0)
    if thin_front is not None:
        L.po_mvs_set_thin_front(mo, thin_front)
    for X, vis in scene.seeds:
        L.po_mvs_add_seed(mo, po.darr(X), len(vis), po.iarr(vis))
    L.po_mvs_refine_seed_patches(mo)
    L.po_mvs_expansion_patches(mo, B, max_rounds, 1)
    rows = []
    for i in range(L.po_mvs_num_slots(mo)):
        pp = L.po_mvs_get_patch(mo, i)
        if pp:
            p = pp.contents
            rows.append((list(p.center[:]), list(p.normalS[:]), p.cams(), p.fitness, p.correlation))
    calls, acc, spec = L.po_mvs_refine_calls(mo), L.po_mvs_num_patches(mo), L.po_mvs_speculative(mo)
    L.po_mvs_destroy(mo)
    S.close()
    return rows, calls, acc, spec
