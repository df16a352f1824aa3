"""SURVEY 8(f) N4, second half: seeds from image features -- FeatureManager::setSeedPatches after its SIFT call
(mvs/featuremanager.cpp:28-99, 118-287) -- include/pais_seed.h against the oracle's restatement (oracle/po_seed.c).

Keypoints and descriptors are synthetic (cv::SIFT is OpenCV non-free and not in the image): 3-D points projected into the
cameras that see them, 128-float descriptors per point with a little per-view noise, outlier keypoints with random
descriptors, and points whose descriptors were swapped between two views' geometry (descriptor matches that the epipolar
filter must reject)."""
import ctypes as C

import numpy as np
import pytest

from tests import common


def synth_features(scene, n_points=160, n_outliers=40, seed=3, dim=128):
    rng = np.random.default_rng(seed)
    cams = scene.cameras
    centre = np.mean([X for X, _ in scene.seeds], axis=0)
    pts = centre + rng.uniform(-0.12, 0.12, size=(n_points, 3))
    base = rng.uniform(0, 255, size=(n_points, dim)).astype(np.float32)
    base[5] = base[4]                                   # two points with the SAME descriptor: ties for the matcher
    xy = [[] for _ in cams]
    desc = [[] for _ in cams]
    for k, X in enumerate(pts):
        for c, cam in enumerate(cams):
            if rng.random() < 0.25:
                continue                                # not detected in this view
            q = cam.rotation @ X + cam.translation
            x = cam.focal[0] * q[0] / q[2] + cam.principle_point[0]
            y = cam.focal[1] * q[1] / q[2] + cam.principle_point[1]
            h, w = cam.image.shape
            if not (1 <= x < w - 1 and 1 <= y < h - 1):
                continue
            noise = rng.normal(0, 0.25, 2)
            if k % 17 == 0 and c == 1:
                noise += np.array([25.0, -18.0])       # right descriptor, wrong place: must fall to the epipolar filter
            xy[c].append((x + noise[0], y + noise[1]))
            desc[c].append(base[k] + rng.normal(0, 1.5, dim).astype(np.float32) * (k not in (4, 5)))
    for c, cam in enumerate(cams):
        h, w = cam.image.shape
        for _ in range(n_outliers):
            xy[c].append((rng.uniform(0, w), rng.uniform(0, h)))
            desc[c].append(rng.uniform(0, 255, dim).astype(np.float32))
        order = rng.permutation(len(xy[c]))
        xy[c] = np.array(xy[c], dtype=np.float32)[order]
        desc[c] = np.array(desc[c], dtype=np.float32)[order]
    return xy, desc


def oracle_features(S, xy, desc, max_dist):
    from oracle import po
    L = po.lib()
    n = len(xy)
    fpp = C.POINTER(C.c_float)
    kn = (C.c_int * n)(*[len(p) for p in xy])
    pxy = (fpp * n)(*[p.ctypes.data_as(fpp) for p in xy])
    pds = (fpp * n)(*[d.ctypes.data_as(fpp) for d in desc])
    feat = C.POINTER(C.c_int)()
    flen = C.c_int(0)
    cen = C.POINTER(C.c_double)()
    k = L.po_seed_features(S.ptr, kn, pxy, pds, desc[0].shape[1], max_dist, C.byref(feat), C.byref(flen), C.byref(cen))
    out, w = [], 0
    for u in range(k):
        m = feat[w]; w += 1
        nodes = [(feat[w + 2 * e], feat[w + 2 * e + 1]) for e in range(m)]
        w += 2 * m
        out.append((nodes, [cen[3 * u + i] for i in range(3)]))
    L.po_seed_free(feat); L.po_seed_free(cen)
    return out


def oracle_match_table(xy, desc):
    from oracle import po
    L = po.lib()
    fpp = C.POINTER(C.c_float)
    table = []
    for i in range(len(xy)):
        for j in range(len(xy)):
            if i == j:
                continue
            tq = (C.c_int * max(len(xy[i]), 1))()
            dd = (C.c_float * max(len(xy[i]), 1))()
            L.po_seed_match(len(xy[i]), desc[i].ctypes.data_as(fpp), len(xy[j]), desc[j].ctypes.data_as(fpp), desc[i].shape[1], tq, dd)
            table += [(i, j, q, tq[q]) for q in range(len(xy[i])) if tq[q] >= 0]
    return table


def _seeds_of(m):
    return [(p.cams(), [tuple(p.imgPoint[k][:]) for k in range(p.num_cam)], list(p.center[:])) for p in m.patches()]


def test_fundamental_matrices(pawn_small):
    from oracle import po
    from pais_mvs_amd import seed
    from pais_mvs_amd.config import readme_config
    from pais_mvs_amd.context import camera_desc
    S = common.oracle_scene(readme_config(), pawn_small)
    L = po.lib()
    keep = []
    descs = [camera_desc(c, False, keep) for c in pawn_small.cameras]
    rng = np.random.default_rng(1)
    centre = np.mean([X for X, _ in pawn_small.seeds], axis=0)
    for a in range(len(descs)):
        for b in range(len(descs)):
            if a == b:
                continue
            got = seed.fundamental(descs[a], descs[b])
            want = np.zeros(9)
            L.po_seed_fundamental(C.byref(S.ptr.contents.cams[a]), C.byref(S.ptr.contents.cams[b]), want.ctypes.data_as(C.POINTER(C.c_double)))
            assert np.array_equal(got.ravel(), want), (a, b)
            # closed form: x_to' F x_from = 0 for the two images of any 3-D point (featuremanager.h:24)
            for _ in range(5):
                X = centre + rng.uniform(-0.1, 0.1, 3)
                xs = []
                for cam in (pawn_small.cameras[a], pawn_small.cameras[b]):
                    q = cam.rotation @ X + cam.translation
                    xs.append(np.array([cam.focal[0] * q[0] / q[2] + cam.principle_point[0], cam.focal[1] * q[1] / q[2] + cam.principle_point[1], 1.0]))
                l = got @ xs[0]
                assert abs(xs[1] @ l) / np.hypot(l[0], l[1]) < 1e-6


def test_seeds_from_matches_equal_the_oracle(pawn_small):
    """Filters, union and seeds on a GPU-less driver (host only), fed with the oracle's match table."""
    from pais_mvs_amd import seed
    from pais_mvs_amd.config import readme_config
    from pais_mvs_amd.mvs import MVS
    cfg = readme_config()
    S = common.oracle_scene(cfg, pawn_small)
    xy, desc = synth_features(pawn_small)
    want = oracle_features(S, xy, desc, 2.0)
    table = oracle_match_table(xy, desc)
    m = MVS(cfg, pawn_small.cameras, device=-1, seed=42)
    n = seed.seeds_from_matches(m, xy, desc, table, 2.0)
    got = _seeds_of(m)
    assert n == len(want) == len(got) and n > 60, (n, len(want))
    sizes = set()
    for (nodes, cen), (cams, pts, c) in zip(want, got):
        assert cams == [a for a, _ in nodes]
        assert pts == [(float(xy[a][f][0]), float(xy[a][f][1])) for a, f in nodes]
        assert c == cen                                  # reCentering: same bits
        sizes.add(len(nodes))
    assert len(sizes) >= 2                               # features of several view counts
    # the wrong-place keypoints never made it into a feature with their true partners: every seed lies near the points
    centre = np.mean([X for X, _ in pawn_small.seeds], axis=0)
    assert max(np.linalg.norm(np.array(c) - centre) for _, _, c in got) < 0.4
    # a stricter epipolar bound keeps fewer, a looser one more
    for md, cmp in ((0.05, lambda a: a < n), (50.0, lambda a: a >= n)):
        m2 = MVS(cfg, pawn_small.cameras, device=-1, seed=42)
        k = seed.seeds_from_matches(m2, xy, desc, table, md)
        assert cmp(k) and k == len(oracle_features(S, xy, desc, md)), (md, k)
        m2.close()
    m.close()


def test_bad_arguments_are_refused(pawn_small):
    from pais_mvs_amd import seed
    from pais_mvs_amd.config import readme_config
    from pais_mvs_amd.mvs import MVS
    xy, desc = synth_features(pawn_small, n_points=10, n_outliers=2)
    m = MVS(readme_config(), pawn_small.cameras, device=-1, seed=42)
    with pytest.raises(RuntimeError):
        seed.seeds_from_matches(m, xy, desc, [(0, 0, 0, 0)], 2.0)           # a camera matched with itself
    with pytest.raises(RuntimeError):
        seed.seeds_from_matches(m, xy, desc, [(0, 1, 10 ** 6, 0)], 2.0)     # keypoint index out of range
    with pytest.raises(RuntimeError):
        seed.set_seed_patches(m, xy, desc, 2.0)                            # the matcher needs the GPU: no CPU path
    with pytest.raises(RuntimeError):
        seed.match(-1, desc[0], desc[1])
    m.close()


@pytest.mark.gpu
def test_descriptor_matching_on_the_gpu_equals_the_oracle(pawn_small):
    from oracle import po
    from pais_mvs_amd import seed
    L = po.lib()
    fpp = C.POINTER(C.c_float)
    rng = np.random.default_rng(9)
    for nq, nt, dim in ((300, 257, 128), (1, 5, 128), (64, 1, 128), (513, 700, 64)):
        q = rng.integers(0, 256, size=(nq, dim)).astype(np.float32)
        t = rng.integers(0, 256, size=(nt, dim)).astype(np.float32)
        k = min(nq, nt) // 2
        t[:k] = q[:k] + rng.normal(0, 2.0, (k, dim)).astype(np.float32)   # true partners
        if nt > 8:
            t[7] = t[6]                                                     # exact duplicates: first minimum wins
        got, gd = seed.match(0, q, t)
        want = (C.c_int * nq)()
        wd = (C.c_float * nq)()
        L.po_seed_match(nq, q.ctypes.data_as(fpp), nt, t.ctypes.data_as(fpp), dim, want, wd)
        assert list(got) == list(want), (nq, nt)
        assert np.array_equal(gd, np.frombuffer(wd, dtype=np.float32)), (nq, nt)   # same float bits
        assert (got >= 0).sum() >= k // 2


@pytest.mark.gpu
def test_set_seed_patches_end_to_end(pawn_small):
    """Keypoints + descriptors -> seeds on the GPU driver == the oracle; the seeds then go through refineSeedPatches."""
    from pais_mvs_amd import seed
    from pais_mvs_amd.config import readme_config
    from pais_mvs_amd.mvs import MVS
    cfg = readme_config()
    S = common.oracle_scene(cfg, pawn_small)
    xy, desc = synth_features(pawn_small)
    want = oracle_features(S, xy, desc, 2.0)
    m = MVS(cfg, pawn_small.cameras, device=0, seed=42)
    n = seed.set_seed_patches(m, xy, desc, 2.0)
    got = _seeds_of(m)
    assert n == len(want) == len(got)
    for (nodes, cen), (cams, pts, c) in zip(want, got):
        assert cams == [a for a, _ in nodes] and c == cen
    m.refineSeedPatches()
    assert m.stats().seeds_refined == n
    m.close()
