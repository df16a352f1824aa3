"""ctypes binding of include/pais_seed.h: seeds from image features (FeatureManager::setSeedPatches after the SIFT call)."""
from __future__ import annotations

import ctypes as C
from typing import List, Sequence, Tuple

import numpy as np

from . import _lib
from .mvs import MVS


class Keypoints(C.Structure):
    _fields_ = [("n", C.c_int32), ("_pad", C.c_int32), ("xy", C.POINTER(C.c_float)), ("desc", C.POINTER(C.c_float))]


class PairMatch(C.Structure):
    _fields_ = [("cam_q", C.c_int32), ("cam_t", C.c_int32), ("q", C.c_int32), ("t", C.c_int32)]


def _bind(L):
    if getattr(L, "_seed_bound", False):
        return L
    fp = C.POINTER(C.c_float)
    L.pais_seed_fundamental.argtypes = [C.POINTER(_lib.CameraDesc), C.POINTER(_lib.CameraDesc), C.POINTER(C.c_double)]
    L.pais_seed_match.argtypes = [C.c_int, C.c_int, fp, C.c_int, fp, C.c_int, C.POINTER(C.c_int32), fp]
    L.pais_mvs_seeds_from_matches.argtypes = [C.c_void_p, C.c_int, C.POINTER(Keypoints), C.c_int, C.POINTER(PairMatch), C.c_double,
                                              C.POINTER(C.c_int)]
    L.pais_mvs_set_seed_patches.argtypes = [C.c_void_p, C.c_int, C.POINTER(Keypoints), C.c_int, C.c_double, C.POINTER(C.c_int)]
    L.pais_seed_last_error.restype = C.c_char_p
    L._seed_bound = True
    return L


def keypoint_array(xy: Sequence[np.ndarray], desc: Sequence[np.ndarray]):
    """-> (ctypes array of pais_keypoints, the float arrays it points into: keep them alive)."""
    keep = []
    arr = (Keypoints * len(xy))()
    for c, (p, d) in enumerate(zip(xy, desc)):
        p = np.ascontiguousarray(p, dtype=np.float32).reshape(-1, 2)
        d = np.ascontiguousarray(d, dtype=np.float32)
        d = d.reshape(len(p), d.size // len(p)) if len(p) else d.reshape(0, 0)   # (a camera without keypoints is fine)
        keep += [p, d]
        arr[c].n = len(p)
        arr[c].xy = p.ctypes.data_as(C.POINTER(C.c_float))
        arr[c].desc = d.ctypes.data_as(C.POINTER(C.c_float))
    return arr, keep


def fundamental(cam_from: _lib.CameraDesc, cam_to: _lib.CameraDesc) -> np.ndarray:
    L = _bind(_lib.load())
    F = np.zeros(9)
    rc = L.pais_seed_fundamental(C.byref(cam_from), C.byref(cam_to), F.ctypes.data_as(C.POINTER(C.c_double)))
    if rc:
        raise RuntimeError(L.pais_seed_last_error().decode())
    return F.reshape(3, 3)


def match(device: int, query: np.ndarray, train: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """BFMatcher(NORM_L2, crossCheck=True).match on the GPU: (train index per query or -1, distance to the nearest)."""
    L = _bind(_lib.load())
    q = np.ascontiguousarray(query, dtype=np.float32)
    t = np.ascontiguousarray(train, dtype=np.float32)
    out = np.full(len(q), -1, dtype=np.int32)
    dist = np.zeros(len(q), dtype=np.float32)
    fp = C.POINTER(C.c_float)
    rc = L.pais_seed_match(device, len(q), q.ctypes.data_as(fp), len(t), t.ctypes.data_as(fp), q.shape[1] if q.ndim == 2 else 0,
                           out.ctypes.data_as(C.POINTER(C.c_int32)), dist.ctypes.data_as(fp))
    if rc:
        raise RuntimeError(L.pais_seed_last_error().decode())
    return out, dist


def seeds_from_matches(m: MVS, xy, desc, matches: List[Tuple[int, int, int, int]], max_dist: float) -> int:
    L = _bind(m.L)
    arr, keep = keypoint_array(xy, desc)
    pm = (PairMatch * max(len(matches), 1))()
    for k, (a, b, q, t) in enumerate(matches):
        pm[k].cam_q, pm[k].cam_t, pm[k].q, pm[k].t = a, b, q, t
    n = C.c_int(0)
    m._check(L.pais_mvs_seeds_from_matches(m.h, len(xy), arr, len(matches), pm, float(max_dist), C.byref(n)), "pais_mvs_seeds_from_matches")
    return n.value


def set_seed_patches(m: MVS, xy, desc, max_dist: float) -> int:
    """FeatureManager::setSeedPatches(cameras, maxDist, mvs) from the keypoints / descriptors on; returns the seeds added."""
    L = _bind(m.L)
    arr, keep = keypoint_array(xy, desc)
    dim = 0
    for p, d in zip(xy, desc):                  # the first camera that has keypoints defines the descriptor length
        npts = len(np.asarray(p).reshape(-1, 2))
        if npts:
            dim = int(np.asarray(d).size // npts)
            break
    if dim <= 0:
        return 0
    n = C.c_int(0)
    m._check(L.pais_mvs_set_seed_patches(m.h, len(xy), arr, dim, float(max_dist), C.byref(n)), "pais_mvs_set_seed_patches")
    return n.value
