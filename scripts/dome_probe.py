"""Why do dome seeds survive or not?  Prints the refine() records of the seeds of a dome rig (GPU)."""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pais_mvs_amd import synth
from pais_mvs_amd.config import readme_config
from pais_mvs_amd.context import Context, make_candidate, normal_to_spherical
w, h, f = (int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])) if len(sys.argv) > 3 else (4096, 3072, 4500.0)
scene = synth.dome_scene(n_cams=128, width=w, height=h, focal=f, n_seeds=60, build_edges=False, device=0)
for vc in (0.7,):
    cfg = readme_config(patchRadius=25, distWeighting=25 / 3.0, reduceNormalRange=4.0, adaptiveGradientEnable=True, visibleCorrelation=vc)
    ctx = Context(cfg, scene.cameras, device=0, seed=42)
    ctx.set_neighbor_radius(0.02)
    cands = []
    for i, (X, vis) in enumerate(scene.seeds):
        n = np.zeros(3)
        for c in vis:
            d = scene.cameras[c].center - X; n += d / np.linalg.norm(d)
        n /= np.linalg.norm(n)
        cands.append(make_candidate(X, n, vis, i, 0))
    res = ctx.refine_batch(cands)
    for r, (X, vis) in list(zip(res, scene.seeds))[:30]:
        print("drop %d K0 %2d K %2d ref %3d lod %2d fit %12.6g corr %.3f runs %d iters %3d depthRange [%.3f %.3f] depth %.3f" % (
            r.dropped, len(vis), r.num_cam, r.ref_cam, r.lod, r.fitness if r.fitness < 1e300 else -1, r.correlation, r.pso_runs, r.pso_iterations,
            r.depthRange[0], r.depthRange[1], r.depth))
    print("kept", sum(1 for r in res if not r.dropped), "of", len(res))
    # cross-check a few final states against the oracle's literal cost
    from tests import common
    from pais_mvs_amd import _lib
    import ctypes as C
    # the oracle reads the edge map of the reference camera only: build those, share zero maps for the rest
    from pais_mvs_amd.camera import sobel_magnitude_normalised
    keep = [r for r in res if not r.dropped][:4] + [r for r in res if r.dropped and r.fitness < 1e300][:2]
    refs = {r.ref_cam for r in keep}
    zeros = [np.zeros(l.shape) for l in scene.cameras[0].pyramid]
    for i, cam in enumerate(scene.cameras):
        cam.edge_pyramid = [sobel_magnitude_normalised(l) for l in cam.pyramid] if i in refs else zeros
    S = common.oracle_scene(cfg, scene)
    if S is not None:
        from oracle import po
        for r in keep:
            p = po.Patch()
            p.numCam = r.num_cam
            for i in range(r.num_cam): p.camIdx[i] = r.cam_idx[i]
            p.refCamIdx = r.ref_cam; p.LOD = r.lod; p.ray[:] = r.ray[:]
            for mode in (False, True):
                S.set_kernel_arithmetic(mode)
                print("   oracle(%s) fitness at the result: %.9g   (GPU said %.9g)" % ("kernel" if mode else "literal", S.fitness(p, [r.normalS[0], r.normalS[1], r.depth]), r.fitness))
    ctx.close()
