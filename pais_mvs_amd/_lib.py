"""ctypes binding of libpais_hip.so (include/pais_hip.h, include/pais_mvs.h).

The product path fails loudly when the HIP library is missing or no GPU is
present: there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os

from . import build as _build

MAX_LEVELS = 16
MAX_VIS = 64


class Config(C.Structure):
    _fields_ = [
        ("cellSize", C.c_int32), ("patchRadius", C.c_int32), ("patchSize", C.c_int32), ("minCamNum", C.c_int32),
        ("textureVariation", C.c_double), ("visibleCorrelation", C.c_double), ("minCorrelation", C.c_double),
        ("maxFitness", C.c_double), ("lodRatio", C.c_double),
        ("minLOD", C.c_int32), ("maxLOD", C.c_int32), ("maxCellPatchNum", C.c_int32), ("_pad0", C.c_int32),
        ("reduceNormalRange", C.c_double),
        ("adaptiveDistanceEnable", C.c_int32), ("adaptiveDifferenceEnable", C.c_int32),
        ("adaptiveGradientEnable", C.c_int32), ("_pad1", C.c_int32),
        ("distWeighting", C.c_double), ("diffWeighting", C.c_double), ("gradientWeighting", C.c_double),
        ("neighborRadius", C.c_double), ("neighborRadiusScalar", C.c_double), ("minRegionRatio", C.c_double),
        ("depthRangeScalar", C.c_double),
        ("particleNum", C.c_int32), ("maxIteration", C.c_int32), ("expansionStrategy", C.c_int32), ("_pad2", C.c_int32),
    ]


class CameraDesc(C.Structure):
    _fields_ = [
        ("focal", C.c_double * 2), ("principle_point", C.c_double * 2), ("rotation", C.c_double * 9),
        ("translation", C.c_double * 3), ("center", C.c_double * 3), ("KR", C.c_double * 9), ("KT", C.c_double * 3),
        ("optical_normal", C.c_double * 3), ("max_lod", C.c_int32), ("_pad", C.c_int32),
        ("level_width", C.c_int32 * MAX_LEVELS), ("level_height", C.c_int32 * MAX_LEVELS),
        ("level_stride", C.c_int64 * MAX_LEVELS),
        ("level_image", C.c_void_p * MAX_LEVELS), ("level_edge", C.c_void_p * MAX_LEVELS),
    ]


class PatchState(C.Structure):
    _fields_ = [("ray", C.c_double * 3), ("ref_cam", C.c_int32), ("lod", C.c_int32), ("num_cam", C.c_int32),
                ("_pad", C.c_int32), ("cam_idx", C.c_int32 * MAX_VIS)]


class Candidate(C.Structure):
    _fields_ = [("center", C.c_double * 3), ("normal", C.c_double * 3), ("normalS", C.c_double * 2),
                ("key", C.c_uint64), ("type", C.c_int32), ("num_cam", C.c_int32), ("cam_idx", C.c_int32 * MAX_VIS)]


class PatchResult(C.Structure):
    _fields_ = [
        ("center", C.c_double * 3), ("normal", C.c_double * 3), ("normalS", C.c_double * 2), ("ray", C.c_double * 3),
        ("depth", C.c_double), ("depthRange", C.c_double * 2), ("fitness", C.c_double), ("priority", C.c_double),
        ("correlation", C.c_double), ("imgPoint", (C.c_double * 2) * MAX_VIS), ("key", C.c_uint64),
        ("type", C.c_int32), ("dropped", C.c_int32), ("num_cam", C.c_int32), ("ref_cam", C.c_int32),
        ("lod", C.c_int32), ("pso_runs", C.c_int32), ("pso_iterations", C.c_int32), ("pso_evals", C.c_int32),
        ("cam_idx", C.c_int32 * MAX_VIS),
        ("stage", C.c_int32), ("before_ref", C.c_int32), ("before_num", C.c_int32), ("after_ref", C.c_int32),
        ("after_num", C.c_int32), ("count", C.c_int32), ("total_cam_num", C.c_int32), ("ncc_tables", C.c_int32),
    ]

    def cams(self):
        return [int(self.cam_idx[i]) for i in range(self.num_cam)]


class KernelStats(C.Structure):
    _fields_ = [("pso_ms", C.c_double), ("begin_ms", C.c_double), ("after_ms", C.c_double),
                ("pso_launches", C.c_int64), ("pso_evals", C.c_int64), ("pso_patches", C.c_int64),
                ("pso_algorithmic_bytes", C.c_double), ("ncc_algorithmic_bytes", C.c_double),
                ("ncc_tables", C.c_int64), ("eval_ms", C.c_double), ("eval_launches", C.c_int64),
                ("eval2_ms", C.c_double), ("eval2_launches", C.c_int64), ("eval2_evals", C.c_int64),
                ("eval2_algorithmic_bytes", C.c_double), ("tile_launches", C.c_int64), ("eval2_busy_ms", C.c_double), ("ring_launches", C.c_int64),
                ("ring_ms", C.c_double), ("ring_evals", C.c_int64), ("ring_algorithmic_bytes", C.c_double), ("ring_fallbacks", C.c_int64)]


_lib = None


def load(build_if_needed: bool = True):
    """Load libpais_hip.so; raises if it is absent (no fallback path exists)."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("PAIS_LIB_PATH") or _build.LIB   # PAIS_LIB_PATH: tuning variants (scripts/sens_variants.py)
    if path != _build.LIB:
        build_if_needed = False
    # several ranks of one job must not race to rebuild the same file: a multi-process launch (torch.distributed.run
    # sets WORLD_SIZE) only ever loads what __graft_entry__.build() / `python -m pais_mvs_amd.build` produced
    if int(os.environ.get("WORLD_SIZE", "1")) > 1 or os.environ.get("PAIS_NO_BUILD"):
        build_if_needed = False
    if build_if_needed and _build.needs_build():
        try:
            _build.build()
        except Exception as e:  # built .so travels to the GPU box; hipcc may be absent there
            if not os.path.exists(path):
                raise RuntimeError("libpais_hip.so is missing and could not be built: %s" % e)
    if not os.path.exists(path):
        raise RuntimeError("libpais_hip.so not found at %s -- run `python -m pais_mvs_amd.build`" % path)
    L = C.CDLL(path)
    L.pais_last_error.restype = C.c_char_p
    for n in ("pais_sizeof_config", "pais_sizeof_camera_desc", "pais_sizeof_candidate", "pais_sizeof_patch_result"):
        getattr(L, n).restype = C.c_size_t
    assert L.pais_sizeof_config() == C.sizeof(Config), (L.pais_sizeof_config(), C.sizeof(Config))
    assert L.pais_sizeof_camera_desc() == C.sizeof(CameraDesc)
    assert L.pais_sizeof_candidate() == C.sizeof(Candidate)
    assert L.pais_sizeof_patch_result() == C.sizeof(PatchResult), (L.pais_sizeof_patch_result(), C.sizeof(PatchResult))
    L.pais_ctx_create.restype = C.c_int
    L.pais_ctx_create.argtypes = [C.POINTER(Config), C.c_int, C.POINTER(CameraDesc), C.c_int, C.c_uint64,
                                  C.POINTER(C.c_void_p)]
    L.pais_ctx_set_config.argtypes = [C.c_void_p, C.POINTER(Config)]
    L.pais_ctx_set_neighbor_radius.argtypes = [C.c_void_p, C.c_double]
    L.pais_ctx_destroy.argtypes = [C.c_void_p]
    L.pais_ctx_destroy.restype = None
    L.pais_ctx_stream.restype = C.c_void_p
    L.pais_ctx_stream.argtypes = [C.c_void_p]
    L.pais_ctx_synchronize.argtypes = [C.c_void_p]
    L.pais_fitness_batch.argtypes = [C.c_void_p, C.c_int, C.POINTER(PatchState), C.c_int, C.POINTER(C.c_int32),
                                     C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.pais_refine_batch.argtypes = [C.c_void_p, C.c_int, C.POINTER(Candidate), C.POINTER(PatchResult)]
    L.pais_refine_batch_device.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    L.pais_refine_batch_begin.argtypes = [C.c_void_p, C.c_int, C.POINTER(Candidate)]
    L.pais_refine_batch_open.argtypes = [C.c_void_p, C.c_int, C.POINTER(Candidate), C.c_int]
    L.pais_refine_batch_enqueue.argtypes = [C.c_void_p, C.c_int]
    L.pais_ctx_set_round_hint.argtypes = [C.c_void_p, C.c_int]
    L.pais_refine_batch_end.argtypes = [C.c_void_p, C.POINTER(C.POINTER(PatchResult))]
    L.pais_ctx_fork_lane.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
    L.pais_get_kernel_stats.argtypes = [C.c_void_p, C.POINTER(KernelStats), C.c_int]
    L.pais_ctx_set_fine_timing.argtypes = [C.c_void_p, C.c_int]
    L.pais_neighbor_count.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.c_double, C.POINTER(C.c_int32), C.POINTER(C.c_double)]
    L.pais_rand31.restype = C.c_uint32
    L.pais_rand31.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32]
    L.pais_child_key.restype = C.c_uint64
    L.pais_child_key.argtypes = [C.c_uint64, C.c_int, C.c_int, C.c_int]
    _lib = L
    return L


def check(rc: int, what: str = ""):
    if rc != 0:
        raise RuntimeError("%s failed (%d): %s" % (what or "pais call", rc, load().pais_last_error().decode()))
