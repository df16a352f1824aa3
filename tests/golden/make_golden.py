"""Generates the committed golden vectors.  Run in the build container only:

    python tests/golden/make_golden.py

* pso_reference_traces.json -- produced by the REFERENCE's own PsoSolver
  (oracle/_ref/libpso_ref.so, compiled from /root/reference/TMVS/pso/*.cpp by
  oracle/Makefile) on analytic objectives with the injected deterministic stream:
  the full sequence of evaluated positions + final gBest / fitness / iteration count.
  The oracle's restatement po_pso_run() must reproduce them bit for bit
  (tests/test_oracle_golden.py); nothing of /root/reference is needed at test time.
* oracle_cost_vectors.json -- regression vectors of the oracle's own cost / refine
  functions on the small synthetic pawn scene (the reference ships no vectors for these:
  SURVEY.md section 4), so that later edits of the oracle cannot drift silently.
"""
import ctypes as C
import json
import math
import os
import struct
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import numpy as np
from oracle import po

DBL_MAX = sys.float_info.max


def hexd(x):
    return struct.pack(">d", float(x)).hex()


def objective(kind, x):
    if kind == "sphere":
        return sum(v * v for v in x)
    if kind == "rosen":
        return sum(100 * (x[i + 1] - x[i] ** 2) ** 2 + (1 - x[i]) ** 2 for i in range(2))
    if kind == "plateau":
        if x[0] > 1.0 or x[1] < -1.5:
            return DBL_MAX
        return math.floor(4 * abs(x[0])) / 4 + abs(x[1] - 0.3) + (x[2] - 1) ** 2
    if kind == "allmax":
        return DBL_MAX
    raise ValueError(kind)


def pso_traces():
    L = po.lib()
    R = po.ref_lib()
    assert R is not None, "oracle/_ref/libpso_ref.so missing (needs /root/reference)"
    out = []
    for kind in ["sphere", "rosen", "plateau", "allmax"]:
        for N, maxIt in [(5, 10), (15, 30), (30, 60)]:
            for key in range(2):
                Lo = [-2.0, -3.0, 0.0]
                Up = [2.0, 1.0, 4.0]
                init = [0.5, -0.5, 2.0]
                rec = []

                def f(pos, obj, rec=rec, kind=kind):
                    x = [pos[0], pos[1], pos[2]]
                    rec.append(x)
                    return objective(kind, x)
                fn = po.FITNESS_FN(f)
                r = po.RngCtx(42, key, 0, 0)
                g = (C.c_double * 3)()
                gf = C.c_double()
                it = C.c_int()
                R.ref_pso_run(3, po.darr(Lo), po.darr(Up), C.cast(fn, C.c_void_p), None, maxIt, N, po.darr(init),
                              C.cast(L.po_rng_cb, C.c_void_p), C.addressof(r), 1, g, C.byref(gf), C.byref(it))
                out.append({"objective": kind, "N": N, "maxIt": maxIt, "key": key, "L": Lo, "U": Up, "init": init,
                            "iterations": it.value, "gBest": [hexd(v) for v in g], "gBestFitness": hexd(gf.value),
                            "draws": r.k, "n_evals": len(rec),
                            # positions of the first 3 and last 3 iterations' evaluations + a checksum of all
                            "head": [[hexd(v) for v in p] for p in rec[:3 * N]],
                            "tail": [[hexd(v) for v in p] for p in rec[-3 * N:]],
                            "xor_all": "%016x" % _xor(rec)})
    return out


def _xor(rec):
    acc = 0
    for i, p in enumerate(rec):
        for v in p:
            acc ^= (int.from_bytes(struct.pack(">d", v), "big") * (2 * i + 1)) & 0xFFFFFFFFFFFFFFFF
    return acc


def cost_vectors():
    from pais_mvs_amd import synth
    from pais_mvs_amd.config import readme_config
    from tests import common
    scene = synth.pawn_scene(width=320, height=240, n_seeds=24)
    out = {"scene": "synth.pawn_scene(width=320, height=240, n_seeds=24)", "image_sha1": [], "cases": []}
    import hashlib
    for c in scene.cameras:
        out["image_sha1"].append(hashlib.sha1(c.image.tobytes()).hexdigest())
    L = po.lib()
    rng = np.random.default_rng(11)
    for weights in [(1, 1, 0), (1, 1, 1), (0, 0, 0)]:
        cfg = readme_config(adaptiveDistanceEnable=bool(weights[0]), adaptiveDifferenceEnable=bool(weights[1]),
                            adaptiveGradientEnable=bool(weights[2]))
        S = common.oracle_scene(cfg, scene)
        for i, (X, vis) in enumerate(scene.seeds[:6]):
            p = S.seed_patch(X, vis, key=i)
            L.po_set_reference_camera(S.ptr, C.byref(p)); L.po_set_depth_and_ray(S.ptr, C.byref(p))
            L.po_set_depth_range(S.ptr, C.byref(p)); L.po_set_lod(S.ptr, C.byref(p))
            for j in range(6):
                if j == 0:
                    pos = [p.normalS[0], p.normalS[1], p.depth]
                elif j == 1:
                    pos = [math.pi - p.normalS[0], p.normalS[1] + math.pi, p.depth]
                elif j == 2:
                    pos = [p.normalS[0], p.normalS[1], p.depth * 0.2]
                else:
                    pos = [p.normalS[0] + rng.normal(0, .2), p.normalS[1] + rng.normal(0, .2), p.depth + rng.normal(0, .01)]
                vals = {}
                for mode in (0, 1):
                    S.set_kernel_arithmetic(bool(mode))
                    vals["kernel" if mode else "literal"] = hexd(S.fitness(p, pos))
                out["cases"].append({"weights": list(weights), "seed": i, "pos": [hexd(v) for v in pos], **vals})
        if weights == (1, 1, 0):
            # whole-refine regression, both arithmetic modes
            for mode in (0, 1):
                S.set_kernel_arithmetic(bool(mode))
                for i, (X, vis) in enumerate(scene.seeds[:4]):
                    p = S.seed_patch(X, vis, key=500 + i)
                    L.po_refine_seed(S.ptr, C.byref(p))
                    out["cases"].append({"refine_seed": i, "mode": "kernel" if mode else "literal", "drop": int(p.drop),
                                         "cams": p.cams(), "LOD": p.LOD, "ref": p.refCamIdx, "runs": p.psoRuns,
                                         "iters": p.psoIters, "center": [hexd(v) for v in p.center],
                                         "normal": [hexd(v) for v in p.normal], "fitness": hexd(p.fitness),
                                         "correlation": hexd(p.correlation), "priority": hexd(p.priority)})
    return out


def schedule_vectors():
    """Regression vectors of the oracle's driver-level functions (LITERAL arithmetic -- the mode that restates the
    reference and does not move with the kernels): the cloud of a short R(B) expansion, the `-f` filter chain on it,
    and MVS::reCentering.  Pins round rule, cell claims, thin front, filters and the SVD solve against silent drift."""
    import hashlib
    from pais_mvs_amd import synth
    from pais_mvs_amd.config import readme_config
    from tests import common
    scene = synth.pawn_scene(width=320, height=240, n_seeds=24)
    cfg = readme_config(particleNum=6, maxIteration=8, maxCellPatchNum=6, minCorrelation=0.6)
    L = po.lib()
    out = {"scene": "synth.pawn_scene(width=320, height=240, n_seeds=24); particleNum 6, maxIteration 8, maxCellPatchNum 6, minCorrelation 0.6",
           "image_sha1": [hashlib.sha1(c.image.tobytes()).hexdigest() for c in scene.cameras], "runs": []}

    def cloud_of(mo):
        pts = []
        for i in range(L.po_mvs_num_slots(mo)):
            pp = L.po_mvs_get_patch(mo, i)
            if pp:
                p = pp.contents
                pts.append((i, list(p.center[:]), list(p.normalS[:]), p.cams(), p.fitness, p.correlation))
        return pts

    def digest(pts):
        h = hashlib.sha1()
        for i, c, ns, cams, fit, corr in pts:
            h.update(struct.pack(">i", i))
            for v in c + ns + [fit, corr]:
                h.update(struct.pack(">d", v))
            h.update(bytes(cams))
        return h.hexdigest()

    for B, thin, rounds in ((1, 64, 30), (8, 64, 20), (8, 0, 20)):
        S = common.oracle_scene(cfg, scene)
        mo = L.po_mvs_create(S.ptr)
        for X, vis in scene.seeds:
            L.po_mvs_add_seed(mo, po.darr(X), len(vis), po.iarr(vis))
        L.po_mvs_refine_seed_patches(mo)
        L.po_mvs_set_thin_front(mo, thin)
        L.po_mvs_expansion_patches(mo, B, rounds, 1)
        pts = cloud_of(mo)
        rec = {"B": B, "thin_front": thin, "max_rounds": rounds, "patches": len(pts), "refine_calls": int(L.po_mvs_refine_calls(mo)),
               "cloud_sha1": digest(pts)}
        if B == 8 and thin == 64:
            # the `-f` chain on this cloud, loaded the way a .mvs file is
            m2 = L.po_mvs_create(S.ptr)
            for i, c, ns, cams, fit, corr in pts:
                L.po_mvs_load_patch(m2, po.darr(c), po.darr(ns), len(cams), po.iarr(cams), fit, corr)
            alive = []
            L.po_mvs_cell_filtering(m2); alive.append(len(cloud_of(m2)))
            L.po_mvs_visibility_filtering(m2); alive.append(len(cloud_of(m2)))
            L.po_mvs_neighbor_cell_filtering(m2, 0.25); alive.append(len(cloud_of(m2)))
            L.po_mvs_neighbor_patch_filtering(m2, 0.25, None); alive.append(len(cloud_of(m2)))
            rec["filter_chain_alive"] = alive
            rec["filter_chain_ids_sha1"] = hashlib.sha1(bytes(str([p[0] for p in cloud_of(m2)]), "ascii")).hexdigest()
            L.po_mvs_destroy(m2)
        L.po_mvs_destroy(mo)
        out["runs"].append(rec)
    # reCentering
    S = common.oracle_scene(cfg, scene)
    rc = []
    for k, (X, vis) in enumerate(scene.seeds[:6]):
        meas = []
        for c in vis:
            cam = scene.cameras[c]
            q = cam.rotation @ np.asarray(X, float) + cam.translation
            meas += [cam.focal[0] * q[0] / q[2] + cam.principle_point[0] + 0.25 * ((k + c) % 3 - 1),
                     cam.focal[1] * q[1] / q[2] + cam.principle_point[1] - 0.5 * ((k * c) % 2)]
        got = (C.c_double * 3)()
        L.po_recenter(S.ptr, len(vis), po.iarr(vis), po.darr(meas), got)
        rc.append({"seed": k, "meas": [hexd(v) for v in meas], "center": [hexd(v) for v in got]})
    out["recenter"] = rc
    return out


if __name__ == "__main__":
    json.dump(pso_traces(), open(os.path.join(HERE, "pso_reference_traces.json"), "w"), indent=0)
    json.dump(cost_vectors(), open(os.path.join(HERE, "oracle_cost_vectors.json"), "w"), indent=0)
    json.dump(schedule_vectors(), open(os.path.join(HERE, "oracle_schedule_vectors.json"), "w"), indent=0)
    print("golden vectors written")
