"""Per-scene GPU context (pais_ctx) and the batch entry points."""
from __future__ import annotations

import ctypes as C
import math
from typing import List, Sequence

import numpy as np

from . import _lib
from .camera import Camera
from .config import MvsConfig


def camera_desc(cam: Camera, with_edges: bool, keep: list) -> "_lib.CameraDesc":
    d = _lib.CameraDesc()
    d.focal[:] = list(map(float, cam.focal))
    d.principle_point[:] = list(map(float, cam.principle_point))
    d.rotation[:] = cam.rotation.ravel().tolist()
    d.translation[:] = cam.translation.tolist()
    d.center[:] = cam.center.tolist()
    d.KR[:] = cam.KR.ravel().tolist()
    d.KT[:] = cam.KT.tolist()
    d.optical_normal[:] = cam.optical_normal.tolist()
    d.max_lod = cam.max_lod
    for l, img in enumerate(cam.pyramid):
        img = np.ascontiguousarray(img, dtype=np.uint8)
        keep.append(img)
        d.level_width[l] = img.shape[1]
        d.level_height[l] = img.shape[0]
        d.level_stride[l] = img.strides[0]
        d.level_image[l] = img.ctypes.data
        if with_edges and cam.edge_pyramid:     # no host edge pyramid: the library evaluates the maps on the fly
            e = np.ascontiguousarray(cam.edge_pyramid[l], dtype=np.float64)
            keep.append(e)
            d.level_edge[l] = e.ctypes.data
    return d


def normal_to_spherical(n) -> List[float]:
    """Utility::normal2Spherical (utility.h:17-22)."""
    return [math.acos(float(n[2])), math.atan2(float(n[1]), float(n[0]))]


class Context:
    """Owns one pais_ctx: scene data resident in HBM on one GPU."""

    def __init__(self, cfg: MvsConfig, cameras: Sequence[Camera], device: int = 0, seed: int = 42):
        self.L = _lib.load()
        self.cfg = cfg
        self.cameras = list(cameras)
        self._keep: list = []
        n = len(self.cameras)
        descs = (_lib.CameraDesc * n)()
        for i, cam in enumerate(self.cameras):
            descs[i] = camera_desc(cam, bool(cfg.adaptiveGradientEnable), self._keep)
        self._c_cfg = cfg.to_c()
        h = C.c_void_p()
        _lib.check(self.L.pais_ctx_create(C.byref(self._c_cfg), n, descs, device, seed, C.byref(h)), "pais_ctx_create")
        self.h = h
        self._keep.clear()   # the library copied everything into HBM

    def close(self):
        if getattr(self, "h", None):
            self.L.pais_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_neighbor_radius(self, r: float):
        _lib.check(self.L.pais_ctx_set_neighbor_radius(self.h, float(r)))

    def fitness_batch(self, states: Sequence["_lib.PatchState"], state_index: Sequence[int], particles) -> np.ndarray:
        ns = len(states)
        arr = (_lib.PatchState * ns)(*states)
        idx = np.ascontiguousarray(state_index, dtype=np.int32)
        pts = np.ascontiguousarray(particles, dtype=np.float64).reshape(-1, 3)
        assert len(idx) == len(pts)
        out = np.empty(len(idx), dtype=np.float64)
        _lib.check(self.L.pais_fitness_batch(self.h, ns, arr, len(idx), idx.ctypes.data_as(C.POINTER(C.c_int32)),
                                             pts.ctypes.data_as(C.POINTER(C.c_double)),
                                             out.ctypes.data_as(C.POINTER(C.c_double))), "pais_fitness_batch")
        return out

    def refine_batch(self, cands: Sequence["_lib.Candidate"]):
        n = len(cands)
        arr = (_lib.Candidate * n)(*cands)
        out = (_lib.PatchResult * n)()
        _lib.check(self.L.pais_refine_batch(self.h, n, arr, out), "pais_refine_batch")
        return out

    def kernel_stats(self, reset: bool = False) -> "_lib.KernelStats":
        st = _lib.KernelStats()
        _lib.check(self.L.pais_get_kernel_stats(self.h, C.byref(st), 1 if reset else 0))
        return st


def make_candidate(center, normal, cam_idx, key: int, ptype: int, normalS=None) -> "_lib.Candidate":
    c = _lib.Candidate()
    c.center[:] = [float(v) for v in center]
    c.normal[:] = [float(v) for v in normal]
    ns = normalS if normalS is not None else normal_to_spherical(normal)
    c.normalS[:] = [float(ns[0]), float(ns[1])]
    c.key = int(key)
    c.type = int(ptype)
    c.num_cam = len(cam_idx)
    for i, v in enumerate(cam_idx):
        c.cam_idx[i] = int(v)
    return c
