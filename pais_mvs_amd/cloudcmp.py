"""Cloud-level comparison of two reconstructions of one scene (north_star: "output point clouds match the reference CPU
run's patch centres / normals within 1e-4 relative L2 and identical visible-camera sets").

Two runs whose arithmetic differs in the last bit accept different patches once a PSO trajectory has branched (DESIGN.md
5.3) and everything downstream of that patch differs with it -- its children, the cells they claim, the order of the
queue -- so clouds are compared as SETS: every patch of A against the nearest patch of B and the other way round.
Plain numpy / scipy; no oracle, no GPU.  Used by tests/test_cloud_parity.py, tests/golden/make_bench_golden.py --literal and
bench.py (config.cloud_vs_literal).
"""
from __future__ import annotations

import numpy as np


def _q(v, p):
    return float(np.quantile(v, p)) if len(v) else 0.0


def cloud_metrics(A: np.ndarray, B: np.ndarray, radius: float, masksA=None, masksB=None) -> dict:
    """A, B: (n, 6) arrays of patch centres and unit normals.  radius: the scene's neighborRadius (mvs.cpp:116-141 -- the
    distance under which the reference itself treats two patches as the same surface element).  masks: per patch the
    visible-camera set as a bit mask (optional).  Distances are Euclidean; `rel` ones are divided by the norm of the centre
    (north_star's "relative L2")."""
    from scipy.spatial import cKDTree
    A = np.asarray(A, np.float64).reshape(-1, 6)
    B = np.asarray(B, np.float64).reshape(-1, 6)
    out = {"n_a": int(len(A)), "n_b": int(len(B)), "count_ratio": float(len(A) / max(len(B), 1)), "neighbor_radius": float(radius)}
    if not len(A) or not len(B):
        return out
    ta, tb = cKDTree(A[:, :3]), cKDTree(B[:, :3])
    for name, X, Y, ty, mx, my in (("a_to_b", A, B, tb, masksA, masksB), ("b_to_a", B, A, ta, masksB, masksA)):
        d, j = ty.query(X[:, :3])
        rel = d / np.maximum(np.linalg.norm(X[:, :3], axis=1), 1e-300)
        # signed: a normal pointing the other way is 180 degrees off, not 0 (patch normals are oriented: towards the cameras)
        dot = np.einsum("ij,ij->i", X[:, 3:], Y[j, 3:])
        ang = np.arccos(np.clip(dot, -1.0, 1.0))
        m = {"dist_over_radius_median": _q(d / radius, 0.5), "dist_over_radius_p95": _q(d / radius, 0.95),
             "dist_over_radius_max": float((d / radius).max()),
             "normal_angle_median_rad": _q(ang, 0.5), "normal_angle_p95_rad": _q(ang, 0.95),
             "flipped_normals": int((dot < 0).sum()),
             "within_radius": float((d <= radius).mean()),
             "identical_centre": float((d == 0).mean()),
             "within_1e-4_rel_centre": float((rel <= 1e-4).mean()),
             "within_1e-4_centre_and_normal": float(((rel <= 1e-4) & (np.linalg.norm(X[:, 3:] - Y[j, 3:], axis=1) <= 1e-4)).mean())}
        if mx is not None and my is not None:
            near = rel <= 1e-4
            same = np.asarray(mx)[near] == np.asarray(my)[j][near]
            m["same_camera_set_among_1e-4_matches"] = float(same.mean()) if near.any() else 1.0
        out[name] = m
    return out


def camera_masks(cam_lists) -> np.ndarray:
    """visible-camera sets -> uint64 bit masks (scenes of <= 64 cameras)"""
    out = np.zeros(len(cam_lists), np.uint64)
    for i, cams in enumerate(cam_lists):
        v = 0
        for c in cams:
            v |= 1 << (int(c) & 63)
        out[i] = v
    return out


def surface_error(scene, cloud: np.ndarray, first_cams, every: int = 1) -> dict:
    """Relative depth error of the patch centres against the scene's analytic surface (pais_mvs_amd.synth), along the ray of
    the first camera of each patch's visible set."""
    cloud = np.asarray(cloud, np.float64).reshape(-1, 6)[::every]
    first_cams = np.asarray(first_cams)[::every]
    errs = np.empty(len(cloud))
    for ci in np.unique(first_cams):
        sel = np.nonzero(first_cams == ci)[0]
        cen = scene.cameras[int(ci)].center
        d = cloud[sel, :3] - cen
        dist = np.linalg.norm(d, axis=1)
        t = scene.obj.intersect(cen, d / dist[:, None])
        errs[sel] = np.abs(t - dist) / dist
    errs = errs[np.isfinite(errs)]
    return {"median": _q(errs, 0.5), "p95": _q(errs, 0.95), "n": int(len(errs))}


def save_compact(path: str, cloud: np.ndarray, cam_lists, meta: dict) -> None:
    """Compact fixture of a cloud: centres / normals as float32 (6e-8 relative: three orders inside the 1e-4 gate), camera
    sets as bit masks, first camera of each set."""
    import json
    cloud = np.asarray(cloud, np.float64).reshape(-1, 6)
    np.savez_compressed(path, centre=cloud[:, :3].astype(np.float32), normal=cloud[:, 3:].astype(np.float32),
                        cams=camera_masks(cam_lists), first_cam=np.array([c[0] for c in cam_lists], np.uint8),
                        meta=np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8))


def load_compact(path: str):
    import json
    z = np.load(path)
    cloud = np.concatenate([z["centre"].astype(np.float64), z["normal"].astype(np.float64)], axis=1)
    return cloud, z["cams"], z["first_cam"], json.loads(bytes(z["meta"]).decode())
