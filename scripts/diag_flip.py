"""Diagnostic: GPU-vs-oracle cost differences along the oracle's own PSO trajectory."""
import sys, os, ctypes as C, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pais_mvs_amd import synth, _lib
from pais_mvs_amd.config import readme_config
from pais_mvs_amd.context import Context, make_candidate
from oracle import po
from tests import common

sc = synth.pawn_scene(width=320, height=240, n_seeds=24)
cfg = readme_config()
S = common.oracle_scene(cfg, sc)
ctx = Context(cfg, sc.cameras, 0, 42)
L = po.lib()
which = [int(a) for a in sys.argv[1:]] or [3]
for si in which:
    X, vis = sc.seeds[si]
    p = S.seed_patch(X, vis, key=1000 + si)
    L.po_set_reference_camera(S.ptr, C.byref(p)); L.po_set_depth_and_ray(S.ptr, C.byref(p))
    L.po_set_depth_range(S.ptr, C.byref(p)); L.po_set_lod(S.ptr, C.byref(p))
    N = cfg.particleNum * 2; maxIt = cfg.maxIteration * 2
    Lo = [0.0, p.normalS[1] - math.pi / 2, p.depthRange[0]]; Up = [math.pi, p.normalS[1] + math.pi / 2, p.depthRange[1]]
    init = [p.normalS[0], p.normalS[1], p.depth]
    fc = po.FitCtx(C.cast(S.ptr, C.c_void_p), C.cast(C.pointer(p), C.c_void_p))
    rc = po.RngCtx(42, p.key, 0, 0)
    cap = 11 * N * maxIt + 2 * maxIt + 16
    trace = (C.c_double * cap)(); tl = C.c_int(0); res = po.PsoResult()
    L.po_pso_run(3, po.darr(Lo), po.darr(Up), C.cast(L.po_fit_cb, C.c_void_p), C.addressof(fc), maxIt, N, po.darr(init),
                 C.cast(L.po_rng_cb, C.c_void_p), C.addressof(rc), 0, C.byref(res), trace, cap, C.byref(tl))
    t = np.array(trace[:tl.value]).reshape(-1, 11 * N + 2)
    print("seed", si, "oracle its", res.iterations, "gBestFit", res.gBestFitness, "K", p.numCam, "LOD", p.LOD, "ref", p.refCamIdx)
    st = _lib.PatchState(); st.ray[:] = p.ray[:]; st.ref_cam = p.refCamIdx; st.lod = p.LOD; st.num_cam = p.numCam
    for k in range(p.numCam): st.cam_idx[k] = p.camIdx[k]
    pos = t[:, :11 * N].reshape(-1, N, 11)[:, :, 0:3].reshape(-1, 3)
    fit = t[:, :11 * N].reshape(-1, N, 11)[:, :, 9].reshape(-1)
    pbf = t[:, :11 * N].reshape(-1, N, 11)[:, :, 10].reshape(-1)
    got = ctx.fitness_batch([st], np.zeros(len(pos), np.int32), pos)
    fin = (fit < 1e300) & (got < 1e300)
    print("  evals", len(pos), "finite", fin.sum(), "mismatch DBL_MAX pattern", ((fit < 1e300) != (got < 1e300)).sum())
    rel = np.abs(got[fin] - fit[fin]) / np.abs(fit[fin])
    print("  rel diff: max %.3e median %.3e" % (rel.max(), np.median(rel)))
    # how close were the oracle's own comparisons (fitness vs previous pBestFitness)?
    f2 = fit.reshape(-1, N); 
    prev_pbf = np.vstack([np.full((1, N), np.nan), pbf.reshape(-1, N)[:-1]])
    gap = np.abs(f2 - prev_pbf) / np.maximum(np.abs(prev_pbf), 1e-300)
    gap = gap[np.isfinite(gap) & (f2 < 1e300) & (prev_pbf < 1e300)]
    print("  smallest |fit - pBestFit|/pBestFit gaps:", np.sort(gap)[:8])
    ng = np.isnan(got).sum(); nf = np.isnan(fit).sum()
    print("  NaNs gpu/oracle", ng, nf)
    # full refine on GPU vs oracle for this seed
    cand = make_candidate(p.center[:], p.normal[:], p.cams(), p.key, 0, normalS=p.normalS[:])
    p2 = S.seed_patch(X, vis, key=1000 + si)
    r = ctx.refine_batch([make_candidate(p2.center[:], p2.normal[:], p2.cams(), p2.key, 0, normalS=p2.normalS[:])])[0]
    L.po_refine_seed(S.ptr, C.byref(p2))
    print("  GPU: runs %d its %d fit %.12g | oracle: runs %d its %d fit %.12g" % (r.pso_runs, r.pso_iterations, r.fitness, p2.psoRuns, p2.psoIters, p2.fitness))
    print("  center rel L2 %.3e normal rel L2 %.3e" % (common.rel_l2(r.center[:], p2.center[:]), common.rel_l2(r.normal[:], p2.normal[:])))
