"""Host mirror of the reference's FileLoader / FileWriter for the formats around the hot path
(include/pais_io.h): config.txt, NVM / NVM2, MVS_V2/V3, PLY, PSR."""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import fields
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from .config import MvsConfig

MAX_VIS = _lib.MAX_VIS


class IoCamera(C.Structure):
    _fields_ = [("file_name", C.c_char * 256), ("focal", C.c_double * 2), ("principle_point", C.c_double * 2),
                ("quaternion", C.c_double * 4), ("center", C.c_double * 3), ("radial_distortion", C.c_double)]


class IoPoint(C.Structure):
    _fields_ = [("center", C.c_double * 3), ("rgb", C.c_uint8 * 3), ("_pad", C.c_uint8), ("num_meas", C.c_int32),
                ("cam_idx", C.c_int32 * MAX_VIS), ("feat_idx", C.c_int32 * MAX_VIS), ("xy", (C.c_double * 2) * MAX_VIS)]


class IoPatch(C.Structure):
    _fields_ = [("center", C.c_double * 3), ("normalS", C.c_double * 2), ("num_cam", C.c_int32),
                ("cam_idx", C.c_int32 * MAX_VIS), ("fitness", C.c_double), ("correlation", C.c_double)]


def _L():
    L = _lib.load()
    if getattr(L, "_io_bound", False):
        return L
    vp = C.c_void_p
    L.pais_io_load_config.argtypes = [C.c_char_p, C.POINTER(_lib.Config)]
    L.pais_io_load_nvm.restype = vp
    L.pais_io_load_nvm.argtypes = [C.c_char_p, C.c_int]
    L.pais_io_load_mvs.restype = vp
    L.pais_io_load_mvs.argtypes = [C.c_char_p, C.POINTER(_lib.Config), C.POINTER(C.c_int)]
    L.pais_io_free.argtypes = [vp]
    L.pais_io_free.restype = None
    for n in ("pais_io_num_cameras", "pais_io_num_points", "pais_io_num_patches", "pais_io_num_truncated"):
        getattr(L, n).argtypes = [vp]
    L.pais_io_get_camera.argtypes = [vp, C.c_int, C.POINTER(IoCamera)]
    L.pais_io_get_point.argtypes = [vp, C.c_int, C.POINTER(IoPoint)]
    L.pais_io_get_patch.argtypes = [vp, C.c_int, C.POINTER(IoPatch)]
    L.pais_io_write_mvs.argtypes = [C.c_char_p, C.POINTER(_lib.Config), C.c_int, C.POINTER(IoCamera), C.c_int, C.POINTER(IoPatch)]
    L.pais_io_write_ply.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_uint8)]
    L.pais_io_write_psr.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.pais_io_sizeof_mvsconfig_disk.restype = C.c_size_t
    L._io_bound = True
    return L


def config_from_c(c: "_lib.Config") -> MvsConfig:
    out = MvsConfig()
    for f in fields(MvsConfig):
        v = getattr(c, f.name)
        setattr(out, f.name, bool(v) if isinstance(getattr(out, f.name), bool) else v)
    return out


def load_config(path: str, base: MvsConfig) -> MvsConfig:
    """FileLoader::loadConfig (fileloader.cpp:474-564): keys present in the file override `base`."""
    c = base.to_c()
    rc = _L().pais_io_load_config(path.encode(), C.byref(c))
    if rc:
        raise IOError("cannot open config file %s" % path)
    return config_from_c(c)


def _check_truncation(h, path):
    n = _L().pais_io_num_truncated(h)
    if n:
        raise IOError("%s: %d camera / measurement list(s) exceed PAIS_MAX_VIS = %d entries; this rig is beyond what the "
                      "refine path tracks per patch" % (path, n, MAX_VIS))


def _collect(h, what):
    L = _L()
    if what == "cameras":
        n, get, T = L.pais_io_num_cameras(h), L.pais_io_get_camera, IoCamera
    elif what == "points":
        n, get, T = L.pais_io_num_points(h), L.pais_io_get_point, IoPoint
    else:
        n, get, T = L.pais_io_num_patches(h), L.pais_io_get_patch, IoPatch
    out = []
    for i in range(n):
        t = T()
        get(h, i, C.byref(t))
        out.append(t)
    return out


def load_nvm(path: str, nvm2: bool = False) -> Tuple[List[IoCamera], List[IoPoint]]:
    """FileLoader::loadNVM / loadNVM2 (fileloader.cpp:251-401)."""
    L = _L()
    h = L.pais_io_load_nvm(path.encode(), 1 if nvm2 else 0)
    if not h:
        raise IOError("cannot open NVM file %s" % path)
    try:
        _check_truncation(h, path)
        return _collect(h, "cameras"), _collect(h, "points")
    finally:
        L.pais_io_free(h)


def load_mvs(path: str):
    """FileLoader::loadMVS (fileloader.cpp:403-472) -> (config or None, cameras, patches)."""
    L = _L()
    c = _lib.Config()
    has = C.c_int(0)
    h = L.pais_io_load_mvs(path.encode(), C.byref(c), C.byref(has))
    if not h:
        raise IOError("cannot open MVS file %s" % path)
    try:
        _check_truncation(h, path)
        return (config_from_c(c) if has.value else None), _collect(h, "cameras"), _collect(h, "patches")
    finally:
        L.pais_io_free(h)


def io_camera(name: str, focal, pp, quaternion, center, radial: float = 0.0) -> IoCamera:
    c = IoCamera()
    c.file_name = name.encode()[:255]
    c.focal[:] = [float(focal[0]), float(focal[1])]
    c.principle_point[:] = [float(pp[0]), float(pp[1])]
    c.quaternion[:] = [float(v) for v in quaternion]
    c.center[:] = [float(v) for v in center]
    c.radial_distortion = float(radial)
    return c


def io_patch(center, normalS, cam_idx, fitness: float, correlation: float) -> IoPatch:
    p = IoPatch()
    p.center[:] = [float(v) for v in center]
    p.normalS[:] = [float(normalS[0]), float(normalS[1])]
    p.num_cam = len(cam_idx)
    for i, v in enumerate(cam_idx):
        p.cam_idx[i] = int(v)
    p.fitness = float(fitness)
    p.correlation = float(correlation)
    return p


def write_mvs(path: str, cfg: MvsConfig, cameras: Sequence[IoCamera], patches: Sequence[IoPatch]):
    """FileWriter::writeMVS (filewriter.cpp:71-102)."""
    ca = (IoCamera * max(len(cameras), 1))(*cameras)
    pa = (IoPatch * max(len(patches), 1))(*patches)
    c = cfg.to_c()
    rc = _L().pais_io_write_mvs(path.encode(), C.byref(c), len(cameras), ca, len(patches), pa)
    if rc:
        raise IOError("cannot write MVS file %s (%d)" % (path, rc))


def write_ply(path: str, centers, normals, bgr=None):
    """FileWriter::writePLY (filewriter.cpp:104-139)."""
    cen = np.ascontiguousarray(centers, dtype=np.float64).reshape(-1, 3)
    nor = np.ascontiguousarray(normals, dtype=np.float64).reshape(-1, 3)
    col = None if bgr is None else np.ascontiguousarray(bgr, dtype=np.uint8).reshape(-1, 3)
    rc = _L().pais_io_write_ply(path.encode(), len(cen), cen.ctypes.data_as(C.POINTER(C.c_double)),
                                nor.ctypes.data_as(C.POINTER(C.c_double)),
                                None if col is None else col.ctypes.data_as(C.POINTER(C.c_uint8)))
    if rc:
        raise IOError("cannot write PLY file %s" % path)


def write_psr(path: str, centers, normals):
    """FileWriter::wirtePSR (filewriter.cpp:141-171)."""
    cen = np.ascontiguousarray(centers, dtype=np.float64).reshape(-1, 3)
    nor = np.ascontiguousarray(normals, dtype=np.float64).reshape(-1, 3)
    rc = _L().pais_io_write_psr(path.encode(), len(cen), cen.ctypes.data_as(C.POINTER(C.c_double)),
                                nor.ctypes.data_as(C.POINTER(C.c_double)))
    if rc:
        raise IOError("cannot write PSR file %s" % path)
