"""Host-side camera + image-pyramid preparation (numpy).

Mirrors what the reference's ``Camera`` constructor produces
(``TMVS/mvs/camera.cpp:45-136``): quaternion -> R, T = -R*C, KR, KT, optical
normal, the gray-level pyramid (level i = level 0 resized by ``lodRatio**i``
with area interpolation) and the min-max-normalised Sobel(ksize=1) magnitude
pyramid.  These arrays are *inputs* of the hot path; the HIP library and the
oracle both receive exactly these bytes (OpenCV's
``resize(INTER_AREA)``/``Sobel`` numerics are not in /root/reference, so the
pyramids are treated as shared inputs rather than something to match).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

MAX_LEVELS = 16


def quaternion_to_rotation(q) -> np.ndarray:
    """camera.cpp:6-35 (q = (w, x, y, z), normalised first)."""
    q = np.asarray(q, dtype=np.float64)
    qq = math.sqrt(float(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]))
    if qq > 0:
        qw, qx, qy, qz = (q / qq).tolist()
    else:
        qw, qx, qy, qz = 1.0, 0.0, 0.0, 0.0
    R = np.empty((3, 3), dtype=np.float64)
    R[0, 0] = qw * qw + qx * qx - qz * qz - qy * qy
    R[0, 1] = 2 * qx * qy - 2 * qz * qw
    R[0, 2] = 2 * qy * qw + 2 * qz * qx
    R[1, 0] = 2 * qx * qy + 2 * qw * qz
    R[1, 1] = qy * qy + qw * qw - qz * qz - qx * qx
    R[1, 2] = 2 * qz * qy - 2 * qx * qw
    R[2, 0] = 2 * qx * qz - 2 * qy * qw
    R[2, 1] = 2 * qy * qz + 2 * qw * qx
    R[2, 2] = qz * qz + qw * qw - qy * qy - qx * qx
    return R


def rotation_to_quaternion(R) -> np.ndarray:
    """Inverse of :func:`quaternion_to_rotation` (for synthetic rigs)."""
    R = np.asarray(R, dtype=np.float64)
    tr = R[0, 0] + R[1, 1] + R[2, 2]
    if tr > 0:
        s = math.sqrt(tr + 1.0) * 2
        w = 0.25 * s
        x = (R[2, 1] - R[1, 2]) / s
        y = (R[0, 2] - R[2, 0]) / s
        z = (R[1, 0] - R[0, 1]) / s
    elif R[0, 0] > R[1, 1] and R[0, 0] > R[2, 2]:
        s = math.sqrt(1.0 + R[0, 0] - R[1, 1] - R[2, 2]) * 2
        w = (R[2, 1] - R[1, 2]) / s
        x = 0.25 * s
        y = (R[0, 1] + R[1, 0]) / s
        z = (R[0, 2] + R[2, 0]) / s
    elif R[1, 1] > R[2, 2]:
        s = math.sqrt(1.0 + R[1, 1] - R[0, 0] - R[2, 2]) * 2
        w = (R[0, 2] - R[2, 0]) / s
        x = (R[0, 1] + R[1, 0]) / s
        y = 0.25 * s
        z = (R[1, 2] + R[2, 1]) / s
    else:
        s = math.sqrt(1.0 + R[2, 2] - R[0, 0] - R[1, 1]) * 2
        w = (R[1, 0] - R[0, 1]) / s
        x = (R[0, 2] + R[2, 0]) / s
        y = (R[1, 2] + R[2, 1]) / s
        z = 0.25 * s
    return np.array([w, x, y, z], dtype=np.float64)


def max_lod(width: int, height: int, lod_ratio: float, cfg_max_lod: int) -> int:
    """camera.cpp:63-64."""
    m = int(math.log(float(max(width, height))) / math.log(1.0 / lod_ratio))
    return min(m, cfg_max_lod)


def _area_matrix(ssize: int, fx: float):
    """Sparse (dsize x ssize) area-interpolation weights for scale factor fx<1.

    Follows OpenCV's area-resize decimation table for a fractional scale
    (scale = 1/fx, destination size = round(ssize*fx))."""
    import scipy.sparse as sp

    dsize = int(round(ssize * fx))  # saturate_cast<int> == cvRound
    dsize = max(dsize, 1)
    scale = 1.0 / fx
    rows, cols, vals = [], [], []
    for dx in range(dsize):
        fsx1 = dx * scale
        fsx2 = fsx1 + scale
        cell = min(scale, ssize - fsx1)
        sx1 = int(math.ceil(fsx1))
        sx2 = int(math.floor(fsx2))
        sx2 = min(sx2, ssize - 1)
        sx1 = min(sx1, sx2)
        if sx1 - fsx1 > 1e-3:
            rows.append(dx); cols.append(sx1 - 1); vals.append((sx1 - fsx1) / cell)
        for sx in range(sx1, sx2):
            rows.append(dx); cols.append(sx); vals.append(1.0 / cell)
        if fsx2 - sx2 > 1e-3:
            rows.append(dx); cols.append(sx2); vals.append(min(min(fsx2 - sx2, 1.0), cell) / cell)
    W = sp.csr_matrix((np.asarray(vals, dtype=np.float64), (rows, cols)), shape=(dsize, ssize))
    # normalise rows exactly to 1 (guards the clamped last cell)
    rs = np.asarray(W.sum(axis=1)).ravel()
    rs[rs == 0] = 1.0
    W = sp.diags(1.0 / rs) @ W
    return W.tocsr(), dsize


def resize_area(img: np.ndarray, fx: float) -> np.ndarray:
    """uint8 area-interpolated down-scale by factor fx (both axes)."""
    h, w = img.shape
    Wx, dw = _area_matrix(w, fx)
    Wy, dh = _area_matrix(h, fx)
    tmp = (Wy @ img.astype(np.float64))          # (dh, w)
    out = (Wx @ tmp.T).T                         # (dh, dw)
    return np.clip(np.rint(out), 0, 255).astype(np.uint8)


def sobel_magnitude_normalised(img: np.ndarray) -> np.ndarray:
    """camera.cpp:72-77: Sobel(ksize=1) x/y (central differences, reflect-101
    border), magnitude, then (m - min) / (max - min)."""
    f = img.astype(np.float64)
    p = np.pad(f, 1, mode="reflect")
    gx = p[1:-1, 2:] - p[1:-1, :-2]
    gy = p[2:, 1:-1] - p[:-2, 1:-1]
    mag = np.sqrt(gx * gx + gy * gy)
    mn, mx = float(mag.min()), float(mag.max())
    if mx > mn:
        return (mag - mn) / (mx - mn)
    return np.zeros_like(mag)


def build_pyramid_gpu(image: np.ndarray, lod_ratio: float, cfg_max_lod: int, build_edges: bool, device: int = 0):
    """Camera::Camera's pyramid and edge maps on the MI355X (include/pais_pyramid.h): returns
    (levels, edges, kernel_ms) with the same arrays resize_area / sobel_magnitude_normalised produce on the host."""
    import ctypes as C
    from . import _lib

    class _Pyr(C.Structure):
        _fields_ = [("max_lod", C.c_int), ("width", C.c_int * 16), ("height", C.c_int * 16),
                    ("image", C.POINTER(C.c_uint8) * 16), ("edge", C.POINTER(C.c_double) * 16), ("kernel_ms", C.c_double)]
    L = _lib.load()
    L.pais_pyramid_build.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_double, C.c_int, C.c_int,
                                     C.POINTER(C.POINTER(_Pyr))]
    L.pais_pyramid_free.argtypes = [C.POINTER(_Pyr)]
    L.pais_pyramid_free.restype = None
    L.pais_pyramid_last_error.restype = C.c_char_p
    img = np.ascontiguousarray(image, dtype=np.uint8)
    h, w = img.shape
    out = C.POINTER(_Pyr)()
    rc = L.pais_pyramid_build(device, img.ctypes.data, w, h, w, float(lod_ratio), int(cfg_max_lod), 1 if build_edges else 0, C.byref(out))
    if rc:
        raise RuntimeError("pais_pyramid_build failed (%d): %s" % (rc, L.pais_pyramid_last_error().decode()))
    try:
        p = out.contents
        levels, edges = [], []
        for l in range(p.max_lod + 1):
            n = p.width[l] * p.height[l]
            levels.append(np.ctypeslib.as_array(p.image[l], shape=(n,)).reshape(p.height[l], p.width[l]).copy())
            if build_edges:
                edges.append(np.ctypeslib.as_array(p.edge[l], shape=(n,)).reshape(p.height[l], p.width[l]).copy())
        return levels, edges, p.kernel_ms
    finally:
        L.pais_pyramid_free(out)


@dataclass
class Camera:
    """Host mirror of PAIS::Camera (mvs/camera.h:15-149)."""

    focal: np.ndarray            # (2,)
    principle_point: np.ndarray  # (2,)
    quaternion: np.ndarray       # (4,)
    center: np.ndarray           # (3,)
    image: np.ndarray            # (H, W) uint8 gray, level 0
    name: str = ""
    rgb: Optional[np.ndarray] = None
    radial_distortion: float = 0.0
    # derived
    rotation: np.ndarray = field(default=None)      # type: ignore
    translation: np.ndarray = field(default=None)   # type: ignore
    KR: np.ndarray = field(default=None)            # type: ignore
    KT: np.ndarray = field(default=None)            # type: ignore
    optical_normal: np.ndarray = field(default=None)  # type: ignore
    max_lod: int = 0
    pyramid: List[np.ndarray] = field(default_factory=list)
    edge_pyramid: List[np.ndarray] = field(default_factory=list)

    def finalize(self, lod_ratio: float, cfg_max_lod: int, build_edges: bool = True, device: Optional[int] = None) -> "Camera":
        """device: build the pyramid / edge maps with the HIP kernels on that GPU (identical arrays) instead of numpy."""
        self.focal = np.asarray(self.focal, dtype=np.float64).reshape(2)
        self.quaternion = np.asarray(self.quaternion, dtype=np.float64).reshape(4)
        self.center = np.asarray(self.center, dtype=np.float64).reshape(3)
        h, w = self.image.shape
        pp = np.asarray(self.principle_point, dtype=np.float64).reshape(2)
        if pp[0] < 0 and pp[1] < 0:  # camera.cpp:101-106
            pp = np.array([float(w >> 1), float(h >> 1)])
        self.principle_point = pp
        R = quaternion_to_rotation(self.quaternion)
        K = np.array([[self.focal[0], 0.0, pp[0]], [0.0, self.focal[1], pp[1]], [0.0, 0.0, 1.0]])
        # sequential k = 0..2 sums like the oracle / OpenCV gemm
        T = np.array([-(R[i, 0] * self.center[0] + R[i, 1] * self.center[1] + R[i, 2] * self.center[2])
                      for i in range(3)])
        self.rotation = R
        self.translation = T
        self.KR = _mm(K, R)
        self.KT = np.array([K[i, 0] * T[0] + K[i, 1] * T[1] + K[i, 2] * T[2] for i in range(3)])
        self.optical_normal = np.array([R[0, i] * 0.0 + R[1, i] * 0.0 + R[2, i] * 1.0 for i in range(3)])
        self.max_lod = max_lod(w, h, lod_ratio, cfg_max_lod)
        if device is not None:
            self.pyramid, self.edge_pyramid, _ = build_pyramid_gpu(self.image, lod_ratio, cfg_max_lod, build_edges, device)
            assert len(self.pyramid) == self.max_lod + 1
            return self
        self.pyramid = [np.ascontiguousarray(self.image, dtype=np.uint8)]
        for i in range(1, self.max_lod + 1):
            self.pyramid.append(np.ascontiguousarray(resize_area(self.pyramid[0], lod_ratio ** i)))
        self.edge_pyramid = []
        if build_edges:
            self.edge_pyramid = [np.ascontiguousarray(sobel_magnitude_normalised(l)) for l in self.pyramid]
        return self

    @property
    def width(self) -> int:
        return int(self.image.shape[1])

    @property
    def height(self) -> int:
        return int(self.image.shape[0])


def _mm(a, b):
    out = np.empty((3, 3), dtype=np.float64)
    for i in range(3):
        for j in range(3):
            s = 0.0
            for k in range(3):
                s += a[i, k] * b[k, j]
            out[i, j] = s
    return out
