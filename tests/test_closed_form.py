"""Closed-form anchors for the OpenCV-dependent pieces of the path (no OpenCV here, no reference vectors).

What the reference's math must satisfy regardless of OpenCV's implementation details:
* cv::fitEllipse on 8 points of an exact ellipse returns that ellipse (axes 2a, 2b), so
  getHomographyRegionRatio (patch.cpp:269-288) is b / a; the window's own 8 points (corners + edge mid-points) under a
  similarity transform give ratio 1 by symmetry;
* Mat::inv() of a 3x3 with an exactly representable inverse returns it (and zeros for a singular matrix, cv::invert);
* the plane-induced homography (patch.cpp:290-330) of a fronto-parallel plane between two cameras that differ by a
  translation parallel to the image plane is a pure pixel translation f * B / Z, between cameras that differ along the
  optical axis a pure scaling about the principal point; at a pyramid level both scale with lodRatio^LOD.
Checked for the oracle's restatement AND for the product's device arithmetic (pais_dev.hpp through the host shim)."""
import ctypes as C
import math

import numpy as np
import pytest

from tests import common
from tests.test_detmath_and_devmath import shim  # noqa: F401  (fixture)


def _fit(points):
    from oracle import po
    L = po.lib()
    xy = np.asarray(points, dtype=np.float32).ravel()
    cx, cy, w, h, ang = (C.c_float(), C.c_float(), C.c_float(), C.c_float(), C.c_float())
    L.po_fit_ellipse(len(points), xy.ctypes.data_as(C.POINTER(C.c_float)), C.byref(cx), C.byref(cy), C.byref(w), C.byref(h), C.byref(ang))
    return cx.value, cy.value, w.value, h.value, ang.value


@pytest.mark.parametrize("a,b,phi,cx,cy", [(30.0, 30.0, 0.0, 100.0, 80.0), (40.0, 10.0, 0.0, 0.0, 0.0), (25.0, 12.5, 0.6, 320.0, 240.0),
                                           (18.0, 3.0, -1.1, 50.5, 700.25), (200.0, 150.0, 2.2, 1000.0, 500.0)])
def test_fit_ellipse_recovers_an_exact_ellipse(a, b, phi, cx, cy):
    ts = np.array([0.1, 0.9, 1.7, 2.4, 3.3, 4.1, 4.9, 5.8])          # 8 points, as getHomographyRegionRatio passes
    pts = [(cx + a * math.cos(t) * math.cos(phi) - b * math.sin(t) * math.sin(phi),
            cy + a * math.cos(t) * math.sin(phi) + b * math.sin(t) * math.cos(phi)) for t in ts]
    fx, fy, w, h, _ = _fit(pts)
    tol = 2e-4 * max(a, abs(cx), abs(cy), 1.0)                        # the points are rounded to float (Point2f)
    assert abs(fx - cx) < tol and abs(fy - cy) < tol
    big, small = max(w, h), min(w, h)
    assert abs(big - 2 * a) < 1e-3 * a and abs(small - 2 * b) < 1e-3 * a
    assert abs(small / big - b / a) < 2e-3


def test_region_ratio_of_the_window_under_similarity_is_one(shim):
    from oracle import po
    from pais_mvs_amd.config import readme_config
    from pais_mvs_amd import synth
    scene = synth.pawn_scene(width=160, height=120, n_seeds=2, build_edges=False)
    S = common.oracle_scene(readme_config(), scene)
    L = po.lib()
    for th, sc, tx, ty in ((0.0, 1.0, 0.0, 0.0), (0.7, 1.0, 5.0, -3.0), (-2.0, 0.8, 100.0, 40.0), (1.2, 2.5, -30.0, 12.0)):
        H = np.array([sc * math.cos(th), -sc * math.sin(th), tx, sc * math.sin(th), sc * math.cos(th), ty, 0.0, 0.0, 1.0])
        pt = np.array([80.3, 60.9])
        Hp = H.ctypes.data_as(C.POINTER(C.c_double))
        want = L.po_region_ratio(S.ptr, pt.ctypes.data_as(C.POINTER(C.c_double)), Hp)
        got = shim.shim_region_ratio(float(pt[0]), float(pt[1]), 15, Hp)
        assert abs(want - 1.0) < 1e-4 and abs(got - 1.0) < 1e-4, (th, sc, want, got)
    # an anisotropic map squeezes the window: the ratio falls towards the squeeze factor (monotone, < 1)
    prev = 1.0
    for k in (0.8, 0.5, 0.25, 0.1):
        H = np.array([1.0, 0, 0, 0, k, 0, 0, 0, 1.0])
        r = shim.shim_region_ratio(80.3, 60.9, 15, H.ctypes.data_as(C.POINTER(C.c_double)))
        assert r < prev and abs(r - k) < 0.15 * k + 0.02, (k, r)
        prev = r


def test_inv3_known_inverses(shim):
    from oracle import po
    L = po.lib()
    cases = [
        (np.diag([2.0, 4.0, 0.5]), np.diag([0.5, 0.25, 2.0])),
        (np.array([[1.0, 2, 3], [0, 1, 4], [5, 6, 0]]), np.array([[-24.0, 18, 5], [20, -15, -4], [-5, 4, 1]])),   # det 1
        (np.array([[2.0, -1, 0], [-1, 2, -1], [0, -1, 2]]), np.array([[3.0, 2, 1], [2, 4, 2], [1, 2, 3]]) / 4.0),
        (np.array([[0.0, 1, 0], [0, 0, 1], [1, 0, 0]]), np.array([[0.0, 0, 1], [1, 0, 0], [0, 1, 0]])),
    ]
    dp = C.POINTER(C.c_double)
    for m, inv in cases:
        for fn in (L.po_inv3, shim.shim_inv3):
            out = np.zeros(9)
            fn(np.ascontiguousarray(m.ravel()).ctypes.data_as(dp), out.ctypes.data_as(dp))
            assert np.array_equal(out.reshape(3, 3), inv), (m, out)
    sing = np.array([[1.0, 2, 3], [2, 4, 6], [0, 1, 1]])                 # det == 0 exactly -> zeros (cv::invert)
    for fn in (L.po_inv3, shim.shim_inv3):
        out = np.ones(9)
        fn(sing.ravel().ctypes.data_as(dp), out.ctypes.data_as(dp))
        assert np.array_equal(out, np.zeros(9))


def _two_cameras(c2, f=500.0, w=64, h=48):
    from pais_mvs_amd.camera import Camera
    cams = []
    for c in ((0.0, 0.0, 0.0), c2):
        cams.append(Camera(focal=np.array([f, f]), principle_point=np.array([w / 2.0, h / 2.0]), quaternion=np.array([1.0, 0, 0, 0]),
                           center=np.array(c), image=np.full((h, w), 128, np.uint8)).finalize(0.8, 15, build_edges=False))
    return cams


@pytest.mark.parametrize("lod", [0, 2])
def test_fronto_parallel_homography_closed_forms(shim, lod):
    from oracle import po
    from pais_mvs_amd.config import readme_config
    L = po.lib()
    f, Z, B = 500.0, 4.0, 0.25
    s = 0.8 ** lod
    dp = C.POINTER(C.c_double)
    n = np.array([0.0, 0.0, -1.0])
    X = np.array([0.1, -0.05, Z])
    for c2, kind in (((B, 0.0, 0.0), "shift"), ((0.0, 0.0, -1.0), "zoom")):
        cams = _two_cameras(c2, f)
        S = po.OracleScene(common.oracle_cfg(readme_config()), cams, seed=1)
        p = S.seed_patch(X, [0, 1], key=0)
        p.refCamIdx = 0
        p.LOD = lod
        H = np.zeros(18)
        L.po_homographies(S.ptr, C.byref(p), X.ctypes.data_as(dp), n.ctypes.data_as(dp), H.ctypes.data_as(dp))
        H0, H1 = H[:9].reshape(3, 3), H[9:].reshape(3, 3)
        assert np.array_equal(H0, np.eye(3))                                  # patch.cpp:317-320
        H1 = H1 / H1[2, 2]
        ppx, ppy = cams[0].principle_point * s
        if kind == "shift":       # same orientation, baseline along x: u' = u - s f B / Z
            want = np.array([[1.0, 0, -s * f * B / Z], [0, 1, 0], [0, 0, 1]])
        else:                     # camera 2 one unit behind camera 1 on the optical axis: scaling Z / (Z + 1) about pp
            k = Z / (Z + 1.0)
            want = np.array([[k, 0, ppx * (1 - k)], [0, k, ppy * (1 - k)], [0, 0, 1]])
        assert np.allclose(H1, want, rtol=0, atol=1e-9), (kind, lod, H1, want)
        # the product's device arithmetic (plane_matrix + inv3 + mul33) on the same cameras
        Hs = np.zeros(9)
        d = -float(X @ n)
        shim.shim_plane_h(C.c_double(d), C.c_double(s), cams[0].KR.ravel().ctypes.data_as(dp), cams[0].KT.ctypes.data_as(dp),
                          cams[1].KR.ravel().ctypes.data_as(dp), cams[1].KT.ctypes.data_as(dp), n.ctypes.data_as(dp), Hs.ctypes.data_as(dp))
        Hs = Hs.reshape(3, 3) / Hs[8]
        assert np.allclose(Hs, want, rtol=0, atol=1e-9)
        # a window pixel mapped by H lands where the 3-D point it sees projects in camera 2
        u = np.array([ppx + 7.0, ppy - 3.0, 1.0])
        ray = np.array([(u[0] / s - cams[0].principle_point[0]) / f, (u[1] / s - cams[0].principle_point[1]) / f, 1.0]) * Z
        q = cams[1].rotation @ ray + cams[1].translation
        proj = np.array([f * q[0] / q[2] + cams[1].principle_point[0], f * q[1] / q[2] + cams[1].principle_point[1]]) * s
        v = H1 @ u
        assert np.allclose(v[:2] / v[2], proj, atol=1e-9)
        S.close()
