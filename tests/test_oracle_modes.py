"""How far apart are the two arithmetic modes of the oracle?  (CPU only)

literal  = the reference's arithmetic: platform libm, sequential sums in its pixel order;
kernel   = what the HIP kernels compute: fdlibm exp/sin/cos, fma taps, wave64 reduction trees.

The COST differs in the last bits only (asserted: <= 1e-12 relative).  Whole refine() runs are a
chaotic map of those bits -- converged PSO particles tie with their personal best within an ulp
(psosolver.cpp:128) -- so individual patches may follow different trajectories; both modes must still
find the same surface: this is where north_star's "centres/normals within 1e-4 relative L2" is checked,
as a statistic, together with the visible-camera sets.
"""
import ctypes as C
import math

import numpy as np

from tests import common


def test_cost_modes_agree_to_rounding(pawn_small):
    from oracle import po
    from pais_mvs_amd.config import readme_config
    L = po.lib()
    rng = np.random.default_rng(5)
    worst = 0.0
    n = 0
    for grad in (False, True):
        cfg = readme_config(adaptiveGradientEnable=grad)
        S = common.oracle_scene(cfg, pawn_small)
        for i, (X, vis) in enumerate(pawn_small.seeds[:10]):
            p = S.seed_patch(X, vis, key=i)
            L.po_set_reference_camera(S.ptr, C.byref(p)); L.po_set_depth_and_ray(S.ptr, C.byref(p))
            L.po_set_depth_range(S.ptr, C.byref(p)); L.po_set_lod(S.ptr, C.byref(p))
            for j in range(10):
                pos = [p.normalS[0] + rng.normal(0, .15), p.normalS[1] + rng.normal(0, .15), p.depth + rng.normal(0, .01)]
                S.set_kernel_arithmetic(False); a = S.fitness(p, pos)
                S.set_kernel_arithmetic(True); b = S.fitness(p, pos)
                if a == common.DBL_MAX or b == common.DBL_MAX:
                    assert a == b
                    continue
                worst = max(worst, abs(a - b) / abs(a)); n += 1
    assert n > 100 and worst <= 1e-12, worst


def test_refine_modes_statistics(pawn_small):
    from oracle import po
    from pais_mvs_amd.config import readme_config
    L = po.lib()
    cfg = readme_config()
    S = common.oracle_scene(cfg, pawn_small)
    S.set_omp(True)
    same_traj = same_cams = both = 0
    dc, dn, dsurf = [], [], []
    obj = pawn_small.obj
    for i, (X, vis) in enumerate(pawn_small.seeds):
        res = []
        for mode in (False, True):
            S.set_kernel_arithmetic(mode)
            p = S.seed_patch(X, vis, key=i)
            L.po_refine_seed(S.ptr, C.byref(p))
            res.append(p)
        a, b = res
        if a.drop or b.drop:
            continue
        both += 1
        same_traj += int(a.psoIters == b.psoIters and a.psoRuns == b.psoRuns)
        same_cams += int(a.cams() == b.cams())
        dc.append(common.rel_l2(b.center[:], a.center[:])); dn.append(common.rel_l2(b.normal[:], a.normal[:]))
        # both answers sit on the true surface equally well
        for p in (a, b):
            cam = pawn_small.cameras[p.refCamIdx]
            d = np.array(p.center[:]) - cam.center
            dist = np.linalg.norm(d)
            t = obj.intersect(cam.center, (d / dist)[None, :])[0]
            dsurf.append(abs(t - dist) / dist)
    assert both >= 12
    dc, dn = np.array(dc), np.array(dn)
    print("\nmodes: %d patches, identical trajectory %d, identical camera sets %d; centre rel-L2 median %.2e max %.2e; "
          "normal rel-L2 median %.2e max %.2e" % (both, same_traj, same_cams, np.median(dc), dc.max(), np.median(dn), dn.max()))
    # identical bits for the patches whose trajectories coincide; the rest differ by PSO convergence noise
    assert same_cams >= 0.8 * both
    assert np.median(dc) <= 1e-4          # north_star tolerance, as a statistic over the batch
    assert dc.max() <= 5e-3 and dn.max() <= 0.2
    assert np.median(dsurf) < 3e-3
