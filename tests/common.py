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


def oracle_reconstruct(cfg, scene, B, max_rounds=0, parallel=True, kernel_arithmetic=True, thin_front=None, seed=42,
                       literal_variant=0, with_cloud=False):
    """Seeds + expansion through the oracle's drivers (po_mvs_refine_seed_patches / po_mvs_expansion_patches).
    Returns (rows for patches_sha1, refine calls, accepted patches, speculative records of the parallel mode)."""
    S = oracle_scene(cfg, scene, seed=seed)
    S.set_kernel_arithmetic(kernel_arithmetic)
    S.set_literal_variant(literal_variant)
    L = po.lib()
    mo = L.po_mvs_create(S.ptr)
    L.po_mvs_set_parallel(mo, 1 if parallel else 0)
    if thin_front is not None:
        L.po_mvs_set_thin_front(mo, thin_front)
    for X, vis in scene.seeds:
        L.po_mvs_add_seed(mo, po.darr(X), len(vis), po.iarr(vis))
    L.po_mvs_refine_seed_patches(mo)
    L.po_mvs_expansion_patches(mo, B, max_rounds, 1)
    rows, cloud = [], []
    for i in range(L.po_mvs_num_slots(mo)):
        pp = L.po_mvs_get_patch(mo, i)
        if pp:
            p = pp.contents
            rows.append((list(p.center[:]), list(p.normalS[:]), p.cams(), p.fitness, p.correlation))
            if with_cloud:
                cloud.append(list(p.center[:]) + list(p.normal[:]))
    calls, acc, spec = L.po_mvs_refine_calls(mo), L.po_mvs_num_patches(mo), L.po_mvs_speculative(mo)
    radius = float(S.ptr.contents.cfg.neighborRadius)
    L.po_mvs_destroy(mo)
    S.close()
    if with_cloud:
        return rows, calls, acc, spec, np.array(cloud, float).reshape(-1, 6), radius
    return rows, calls, acc, spec


# ---------------------------------------------------------------------------------------------------------------------
# candidate-level helpers shared by the literal gate, its control, and the cloud-level comparison
# ---------------------------------------------------------------------------------------------------------------------
def oracle_patch_from_candidate(c) -> "po.Patch":
    """The constructed patch state of a product candidate on the oracle side (what tests/test_scheduler_cpu.py does)."""
    p = po.Patch()
    p.id = -1; p.refCamIdx = -1; p.LOD = -1
    p.fitness = DBL_MAX; p.priority = DBL_MAX
    p.type = c.type
    p.center[:] = c.center[:]
    p.normal[:] = c.normal[:]
    p.normalS[:] = c.normalS[:]
    p.numCam = c.num_cam
    for i in range(c.num_cam):
        p.camIdx[i] = c.cam_idx[i]
    p.key = c.key
    return p


def oracle_refine_patch(S, p, is_seed):
    """refine() of one constructed patch as the two drivers call it (mvs.cpp:214-215, 573-574)."""
    L = po.lib()
    if is_seed:
        L.po_refine_seed(S.ptr, C.byref(p))
    else:
        if p.numCam < S.cfg.minCamNum:
            p.drop = 1          # expandVisibleCamera :758-760
        L.po_refine(S.ptr, C.byref(p))
        L.po_remove_invisible_camera(S.ptr, C.byref(p))
    return p


def record_from_oracle_patch(p, r=None):
    r = r if r is not None else _lib.PatchResult()
    r.center[:] = p.center[:]; r.normal[:] = p.normal[:]; r.normalS[:] = p.normalS[:]
    r.ray[:] = p.ray[:]; r.depth = p.depth; r.depthRange[:] = p.depthRange[:]
    r.fitness = p.fitness; r.priority = p.priority; r.correlation = p.correlation
    for i in range(p.numCam):
        r.imgPoint[i][0] = p.imgPoint[i][0]; r.imgPoint[i][1] = p.imgPoint[i][1]
        r.cam_idx[i] = p.camIdx[i]
    r.key = p.key; r.type = p.type; r.dropped = p.drop; r.num_cam = p.numCam
    r.ref_cam = p.refCamIdx; r.lod = p.LOD
    r.pso_runs = p.psoRuns; r.pso_iterations = p.psoIters; r.pso_evals = p.psoEvals
    return r


def oracle_refine_many(S, cands, is_seed, threads=None):
    """refine() of many product candidates by the oracle, one candidate per host thread (ctypes releases the GIL;
    refine() is a pure function of (scene, candidate)) -> oracle patches in candidate order."""
    import os
    from concurrent.futures import ThreadPoolExecutor
    pats = [oracle_patch_from_candidate(c) for c in cands]
    th = threads or min(64, os.cpu_count() or 1)
    with ThreadPoolExecutor(th) as ex:
        list(ex.map(lambda p: oracle_refine_patch(S, p, is_seed), pats))
    return pats


def copy_struct(x):
    y = type(x)()
    C.memmove(C.byref(y), C.byref(x), C.sizeof(type(x)))
    return y


def cpu_workload_candidates(cfg, scene, B=4096, rounds=(5, 25), per_round=160, kernel_arithmetic=True, refine=None):
    """bench.py's reconstruction driven round by round through the product's stepwise scheduler WITHOUT a GPU: the records come
    from the oracle (kernel arithmetic = the HIP path's records bit for bit, tests/test_gpu_parity.py) or from `refine`
    (cands, n, is_seed) -> PatchResult array.  Returns (kept candidates of rounds[0]..rounds[1], every n/per_round-th of a
    round -- the sample of test_literal_gate_on_expansion_candidates_of_the_bench_workload --, their records, rounds run)."""
    from pais_mvs_amd.mvs import MVS
    S = oracle_scene(cfg, scene)
    S.set_kernel_arithmetic(kernel_arithmetic)

    def by_oracle(cands, n, is_seed):
        out = (_lib.PatchResult * max(n, 1))()
        for k, p in enumerate(oracle_refine_many(S, [cands[i] for i in range(n)], is_seed)):
            record_from_oracle_patch(p, out[k])
        return out

    refine = refine or by_oracle
    m = MVS(cfg, scene.cameras, device=-1, seed=42)
    for X, vis in scene.seeds:
        m.add_seed(X, vis)
    cands, n = m.seed_begin()
    S.ptr.contents.cfg.neighborRadius = m.neighbor_radius()
    m.seed_commit(refine(cands, n, True), n)
    m.expansion_begin()
    S.ptr.contents.cfg.neighborRadius = m.neighbor_radius()
    kept_c, kept_r = [], []
    rnd = 0
    while True:
        done, cands, n = m.round_begin(B)
        if done:
            break
        out = refine(cands, n, False)
        if n and rounds[0] <= rnd <= rounds[1]:
            for i in range(0, n, max(1, n // per_round)):
                kept_c.append(copy_struct(cands[i]))
                kept_r.append(copy_struct(out[i]))
        m.round_commit(out, n)
        rnd += 1
    m.expansion_end()
    m.close()
    S.close()
    return kept_c, kept_r, rnd


def trajectory_split(a_pats, b_pats):
    """Two lists of oracle patches refined from the same candidates in two arithmetics -> statistics of the comparison
    per candidate: same discrete PSO trajectory (po_patch::psoSig, runs, iterations) or branched; centre / normal distances;
    discrete outputs (drop, camera set, reference camera, LOD)."""
    n = 0
    same, branched = [], []
    set_same = set_br = 0
    for a, b in zip(a_pats, b_pats):
        br = not (a.psoSig == b.psoSig and a.psoRuns == b.psoRuns and a.psoIters == b.psoIters)
        if bool(a.drop) != bool(b.drop) or (not a.drop and (a.cams() != b.cams() or a.refCamIdx != b.refCamIdx or a.LOD != b.LOD)):
            set_same += int(not br)
            set_br += int(br)
        if a.drop or b.drop:
            continue
        n += 1
        d = (rel_l2(b.center[:], a.center[:]), rel_l2(b.normal[:], a.normal[:]))
        (branched if br else same).append(d)
    same, branched = np.array(same).reshape(-1, 2), np.array(branched).reshape(-1, 2)
    q = lambda v, p: float(np.quantile(v, p)) if len(v) else 0.0
    return {"n": n, "same_trajectory": len(same), "branched": len(branched),
            "branched_fraction": len(branched) / max(n, 1),
            "same_centre_max": float(same[:, 0].max()) if len(same) else 0.0,
            "same_normal_max": float(same[:, 1].max()) if len(same) else 0.0,
            "branched_centre_median": q(branched[:, 0], 0.5), "branched_centre_p95": q(branched[:, 0], 0.95),
            "branched_centre_max": float(branched[:, 0].max()) if len(branched) else 0.0,
            "branched_normal_median": q(branched[:, 1], 0.5), "branched_normal_p95": q(branched[:, 1], 0.95),
            "branched_normal_max": float(branched[:, 1].max()) if len(branched) else 0.0,
            "beyond_1e-4": int(((branched[:, 0] > 1e-4) | (branched[:, 1] > 1e-4)).sum()) if len(branched) else 0,
            "set_mismatch_on_the_same_trajectory": set_same, "set_mismatch_among_branched": set_br}


LITERAL_VARIANTS = {0: "literal (the reference's statements)",
                    1: "literal, final quotient as fitness * (1 / sumWeight): one rounding perturbed",
                    2: "literal, window sums accumulated y outer / x inner",
                    4: "literal, fused multiply-adds as a contracting compiler emits them",
                    6: "literal, y-outer sums AND fused multiply-adds (another loop order and another compiler)"}
CONTROL_VARIANTS = (1, 2, 4, 6)


def literal_control(cfg, scene, kept_c, threads=None, is_seed=False):
    """The candidates refined by the oracle in literal arithmetic, in its control variants and in kernel arithmetic
    -> {"kernel": split(literal, kernel), "variant_1": split(literal, variant 1), ...}."""
    S = oracle_scene(cfg, scene)
    runs = {}
    for v in (0,) + CONTROL_VARIANTS:
        S.set_kernel_arithmetic(False)
        S.set_literal_variant(v)
        runs[v] = oracle_refine_many(S, kept_c, is_seed, threads)
    S.set_literal_variant(0)
    S.set_kernel_arithmetic(True)
    runs["kernel"] = oracle_refine_many(S, kept_c, is_seed, threads)
    # the HIP path's PAIS_ARITH=literal (pais_literal.hpp): the cost in the reference's statements and summation ORDER, with the
    # deterministic exp / sin / cos; everything outside the cost as in kernel arithmetic
    S.set_cost_literal(True)
    runs["cost_literal"] = oracle_refine_many(S, kept_c, is_seed, threads)
    S.set_cost_literal(False)
    S.close()
    out = {"kernel": trajectory_split(runs[0], runs["kernel"]), "cost_literal": trajectory_split(runs[0], runs["cost_literal"])}
    for v in CONTROL_VARIANTS:
        out["variant_%d" % v] = trajectory_split(runs[0], runs[v])
    # which candidates branch: is it the same set under every perturbation?
    sig = lambda p: (p.psoSig, p.psoRuns, p.psoIters, p.drop)
    br = {k: {i for i, (a, b) in enumerate(zip(runs[0], runs[k])) if sig(a) != sig(b)} for k in ("kernel",) + CONTROL_VARIANTS}
    union_ctl = set().union(*[br[v] for v in CONTROL_VARIANTS])
    out["overlap"] = {"all_candidates": len(kept_c), "kernel_branched": len(br["kernel"]), "control_union": len(union_ctl),
                      "kernel_and_control_union": len(br["kernel"] & union_ctl),
                      "kernel_only": len(br["kernel"] - union_ctl)}
    return out, runs
