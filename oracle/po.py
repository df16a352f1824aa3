"""ctypes binding of oracle/libpais_oracle.so (and oracle/_ref/libpso_ref.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, by ``__graft_entry__.smoke()``
and by ``bench.py``'s ``cpu_baseline`` leg -- never by the product package.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import List, Optional, Sequence

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
MAX_LEVELS = 16
MAX_VIS = 64


class Config(C.Structure):
    _fields_ = [
        ("cellSize", C.c_int), ("patchRadius", C.c_int), ("patchSize", C.c_int), ("minCamNum", C.c_int),
        ("textureVariation", C.c_double), ("visibleCorrelation", C.c_double), ("minCorrelation", C.c_double),
        ("maxFitness", C.c_double), ("lodRatio", C.c_double),
        ("minLOD", C.c_int), ("maxLOD", C.c_int), ("maxCellPatchNum", C.c_int),
        ("reduceNormalRange", C.c_double),
        ("adaptiveDistanceEnable", C.c_int), ("adaptiveDifferenceEnable", C.c_int), ("adaptiveGradientEnable", C.c_int),
        ("distWeighting", C.c_double), ("diffWeighting", C.c_double), ("gradientWeighting", C.c_double),
        ("neighborRadius", C.c_double), ("neighborRadiusScalar", C.c_double), ("minRegionRatio", C.c_double),
        ("depthRangeScalar", C.c_double),
        ("particleNum", C.c_int), ("maxIteration", C.c_int), ("expansionStrategy", C.c_int),
    ]


class CameraS(C.Structure):
    _fields_ = [
        ("focal", C.c_double * 2), ("pp", C.c_double * 2), ("R", C.c_double * 9), ("T", C.c_double * 3),
        ("C", C.c_double * 3), ("KR", C.c_double * 9), ("KT", C.c_double * 3), ("optN", C.c_double * 3),
        ("maxLOD", C.c_int),
        ("width", C.c_int * MAX_LEVELS), ("height", C.c_int * MAX_LEVELS),
        ("img", C.c_void_p * MAX_LEVELS), ("edge", C.c_void_p * MAX_LEVELS),
    ]


class SceneS(C.Structure):
    _fields_ = [
        ("cfg", Config), ("numCams", C.c_int), ("cams", C.POINTER(CameraS)), ("gauss", C.POINTER(C.c_double)),
        ("lodScale", C.c_double * MAX_LEVELS), ("seed", C.c_uint64), ("ompParticles", C.c_int),
        ("detMath", C.c_int), ("treeSum", C.c_int), ("windowPerParticle", C.c_int), ("literalVariant", C.c_int),
        ("costLiteral", C.c_int),
    ]


class Patch(C.Structure):
    _fields_ = [
        ("id", C.c_int), ("type", C.c_int), ("drop", C.c_int), ("expanded", C.c_int),
        ("center", C.c_double * 3), ("numCam", C.c_int), ("camIdx", C.c_int * MAX_VIS), ("refCamIdx", C.c_int),
        ("normalS", C.c_double * 2), ("normal", C.c_double * 3), ("ray", C.c_double * 3), ("depth", C.c_double),
        ("depthRange", C.c_double * 2), ("LOD", C.c_int), ("imgPoint", (C.c_double * 2) * MAX_VIS),
        ("fitness", C.c_double), ("priority", C.c_double), ("correlation", C.c_double),
        ("corrTable", C.c_double * (MAX_VIS * MAX_VIS)),
        ("key", C.c_uint64), ("psoRuns", C.c_int), ("psoIters", C.c_int), ("psoEvals", C.c_int), ("pad0", C.c_int),
        ("psoSig", C.c_uint64),
    ]

    def cams(self) -> List[int]:
        return [int(self.camIdx[i]) for i in range(self.numCam)]


class PsoResult(C.Structure):
    _fields_ = [("gBest", C.c_double * 3), ("gBestFitness", C.c_double), ("iterations", C.c_int), ("evals", C.c_int),
                ("gbestSig", C.c_uint64)]


class FitCtx(C.Structure):
    _fields_ = [("s", C.c_void_p), ("p", C.c_void_p)]


class RngCtx(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("key", C.c_uint64), ("run", C.c_uint32), ("k", C.c_uint32)]


FITNESS_FN = C.CFUNCTYPE(C.c_double, C.POINTER(C.c_double), C.c_void_p)
RAND_FN = C.CFUNCTYPE(C.c_uint32, C.c_void_p)

_lib = None
_ref = None


def build(force: bool = False) -> None:
    """Compile the oracle (and oracle/_ref when /root/reference exists)."""
    so = os.path.join(HERE, "libpais_oracle.so")
    src = os.path.join(HERE, "pais_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", HERE, "libpais_oracle.so"], stdout=subprocess.DEVNULL)
    if os.path.isdir("/root/reference/TMVS/pso"):
        subprocess.check_call(["make", "-C", HERE, "ref"], stdout=subprocess.DEVNULL)


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(os.path.join(HERE, "libpais_oracle.so"))
    L.po_rand31.restype = C.c_uint32
    L.po_rand31.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32]
    L.po_child_key.restype = C.c_uint64
    L.po_child_key.argtypes = [C.c_uint64, C.c_int, C.c_int, C.c_int]
    L.po_config_defaults.argtypes = [C.POINTER(Config)]
    L.po_config_readme.argtypes = [C.POINTER(Config)]
    L.po_camera_init.argtypes = [C.POINTER(CameraS), C.POINTER(C.c_double), C.POINTER(C.c_double),
                                 C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.po_camera_max_lod.restype = C.c_int
    L.po_camera_max_lod.argtypes = [C.c_int, C.c_int, C.c_double, C.c_int]
    L.po_scene_create.restype = C.POINTER(SceneS)
    L.po_scene_create.argtypes = [C.POINTER(Config), C.c_int, C.POINTER(CameraS), C.c_uint64]
    L.po_scene_destroy.argtypes = [C.POINTER(SceneS)]
    L.po_init_gauss.argtypes = [C.POINTER(Config), C.POINTER(C.c_double)]
    L.po_spherical2normal.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.po_normal2spherical.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.po_project.restype = C.c_int
    L.po_project.argtypes = [C.POINTER(SceneS), C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int]
    L.po_inv3.restype = C.c_int
    L.po_inv3.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.po_homographies.argtypes = [C.POINTER(SceneS), C.POINTER(Patch), C.POINTER(C.c_double),
                                  C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.po_region_ratio.restype = C.c_double
    L.po_region_ratio.argtypes = [C.POINTER(SceneS), C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.po_fit_ellipse.argtypes = [C.c_int, C.POINTER(C.c_float)] + [C.POINTER(C.c_float)] * 5
    L.po_get_fitness.restype = C.c_double
    L.po_get_fitness.argtypes = [C.POINTER(SceneS), C.POINTER(Patch), C.POINTER(C.c_double)]
    L.po_pso_run.argtypes = [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_void_p, C.c_void_p,
                             C.c_int, C.c_int, C.POINTER(C.c_double), C.c_void_p, C.c_void_p, C.c_int,
                             C.POINTER(PsoResult), C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_int)]
    L.po_patch_init_seed.argtypes = [C.POINTER(SceneS), C.POINTER(Patch), C.POINTER(C.c_double), C.c_int,
                                     C.POINTER(C.c_int), C.c_uint64]
    L.po_patch_init_expand.argtypes = [C.POINTER(SceneS), C.POINTER(Patch), C.POINTER(C.c_double),
                                       C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_int), C.c_uint64]
    for name in ("po_set_estimated_normal", "po_set_reference_camera", "po_set_depth_and_ray", "po_set_depth_range",
                 "po_set_lod", "po_set_priority", "po_set_image_point", "po_pso_optimization",
                 "po_remove_invisible_camera", "po_expand_visible_camera", "po_refine", "po_refine_seed"):
        getattr(L, name).argtypes = [C.POINTER(SceneS), C.POINTER(Patch)]
    L.po_set_correlation_table.argtypes = [C.POINTER(SceneS), C.POINTER(Patch), C.POINTER(C.c_double)]
    L.po_is_neighbor.restype = C.c_int
    L.po_is_neighbor.argtypes = [C.POINTER(SceneS), C.POINTER(Patch), C.POINTER(Patch)]
    L.po_expand_candidates_parallel.argtypes = [C.POINTER(SceneS), C.POINTER(Patch), C.c_int, C.POINTER(C.c_double),
                                                C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                                C.POINTER(C.c_uint64)]
    L.po_expand_candidates_parallel.restype = None
    L.po_expand_candidate.argtypes = [C.POINTER(SceneS), C.POINTER(Patch), C.POINTER(C.c_double),
                                      C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_int), C.c_uint64]
    L.po_mvs_create.restype = C.c_void_p
    L.po_mvs_create.argtypes = [C.POINTER(SceneS)]
    L.po_mvs_destroy.argtypes = [C.c_void_p]
    L.po_mvs_add_seed.restype = C.c_int
    L.po_mvs_add_seed.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_int)]
    L.po_mvs_set_neighbor_radius.argtypes = [C.c_void_p]
    L.po_mvs_refine_seed_patches.argtypes = [C.c_void_p]
    L.po_mvs_set_thin_front.argtypes = [C.c_void_p, C.c_int]
    L.po_mvs_set_parallel.argtypes = [C.c_void_p, C.c_int]
    L.po_mvs_speculative.restype = C.c_long
    L.po_mvs_speculative.argtypes = [C.c_void_p]
    L.po_recenter.argtypes = [C.POINTER(SceneS), C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_double)]
    # po_seed.c: FeatureManager::setSeedPatches after the SIFT call
    fpp = C.POINTER(C.c_float)
    L.po_seed_fundamental.argtypes = [C.POINTER(CameraS), C.POINTER(CameraS), C.POINTER(C.c_double)]
    L.po_seed_fundamental.restype = None
    L.po_seed_match.argtypes = [C.c_int, fpp, C.c_int, fpp, C.c_int, C.POINTER(C.c_int), fpp]
    L.po_seed_match.restype = None
    L.po_seed_features.argtypes = [C.POINTER(SceneS), C.POINTER(C.c_int), C.POINTER(fpp), C.POINTER(fpp), C.c_int, C.c_double,
                                   C.POINTER(C.POINTER(C.c_int)), C.POINTER(C.c_int), C.POINTER(C.POINTER(C.c_double))]
    L.po_seed_features.restype = C.c_int
    L.po_seed_free.argtypes = [C.c_void_p]
    L.po_seed_free.restype = None
    L.po_mvs_load_patch.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_int), C.c_double, C.c_double]
    L.po_mvs_cell_filtering.argtypes = [C.c_void_p]
    L.po_mvs_visibility_filtering.argtypes = [C.c_void_p]
    L.po_mvs_neighbor_cell_filtering.argtypes = [C.c_void_p, C.c_double]
    L.po_mvs_neighbor_patch_filtering.argtypes = [C.c_void_p, C.c_double, C.POINTER(C.c_int)]
    L.po_mvs_expansion_patches.restype = C.c_long
    L.po_mvs_expansion_patches.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
    L.po_mvs_num_patches.restype = C.c_int
    L.po_mvs_num_patches.argtypes = [C.c_void_p]
    L.po_mvs_num_slots.restype = C.c_int
    L.po_mvs_num_slots.argtypes = [C.c_void_p]
    L.po_mvs_get_patch.restype = C.POINTER(Patch)
    L.po_mvs_get_patch.argtypes = [C.c_void_p, C.c_int]
    L.po_mvs_refine_calls.restype = C.c_long
    L.po_mvs_refine_calls.argtypes = [C.c_void_p]
    L.po_mvs_fitness_evals.restype = C.c_long
    L.po_mvs_fitness_evals.argtypes = [C.c_void_p]
    L.po_runtime_filtering.restype = C.c_int
    L.po_runtime_filtering.argtypes = [C.c_void_p, C.POINTER(Patch)]
    L.po_expansion_center.argtypes = [C.POINTER(SceneS), C.c_int, C.POINTER(Patch), C.c_int, C.c_int,
                                      C.POINTER(C.c_double)]
    for n in ("po_exp_det", "po_exp_poly", "po_sin_det", "po_cos_det"):
        getattr(L, n).restype = C.c_double
        getattr(L, n).argtypes = [C.c_double]
    L.po_sizeof_patch.restype = C.c_size_t
    L.po_sizeof_config.restype = C.c_size_t
    L.po_sizeof_camera.restype = C.c_size_t
    assert L.po_sizeof_patch() == C.sizeof(Patch), (L.po_sizeof_patch(), C.sizeof(Patch))
    assert L.po_sizeof_config() == C.sizeof(Config)
    assert L.po_sizeof_camera() == C.sizeof(CameraS)
    _lib = L
    return L


def ref_lib():
    """The reference's own PsoSolver (oracle/_ref/libpso_ref.so); None if not built."""
    global _ref
    if _ref is not None:
        return _ref
    p = os.path.join(HERE, "_ref", "libpso_ref.so")
    if not os.path.exists(p):
        if os.path.isdir("/root/reference/TMVS/pso"):
            build()
        if not os.path.exists(p):
            return None
    R = C.CDLL(p)
    R.ref_pso_run.restype = C.c_int
    R.ref_pso_run.argtypes = [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_void_p, C.c_void_p,
                              C.c_int, C.c_int, C.POINTER(C.c_double), C.c_void_p, C.c_void_p, C.c_int,
                              C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int)]
    R.ref_pso_rand_max.restype = C.c_int
    _ref = R
    return R


def dptr(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def darr(vals: Sequence[float]):
    return (C.c_double * len(vals))(*[float(v) for v in vals])


def iarr(vals: Sequence[int]):
    return (C.c_int * len(vals))(*[int(v) for v in vals])


def config_readme(**over) -> Config:
    c = Config()
    lib().po_config_readme(C.byref(c))
    for k, v in over.items():
        setattr(c, k, v)
    c.patchSize = 2 * c.patchRadius + 1
    return c


def config_defaults(**over) -> Config:
    c = Config()
    lib().po_config_defaults(C.byref(c))
    for k, v in over.items():
        setattr(c, k, v)
    c.patchSize = 2 * c.patchRadius + 1
    return c


class OracleScene:
    """Owns a po_scene built from host cameras (pais_mvs_amd.camera.Camera-like objects)."""

    def __init__(self, cfg: Config, cameras, seed: int = 42):
        L = lib()
        self.cfg = cfg
        self.cameras = cameras
        n = len(cameras)
        self._cams = (CameraS * n)()
        self._keep = []
        for i, cam in enumerate(cameras):
            cs = self._cams[i]
            L.po_camera_init(C.byref(cs), darr(cam.focal), darr(cam.principle_point), darr(cam.quaternion),
                             darr(cam.center))
            cs.maxLOD = L.po_camera_max_lod(cam.width, cam.height, cfg.lodRatio, cfg.maxLOD)
            assert cs.maxLOD == cam.max_lod, (cs.maxLOD, cam.max_lod)
            for l, img in enumerate(cam.pyramid):
                img = np.ascontiguousarray(img, dtype=np.uint8)
                self._keep.append(img)
                cs.width[l] = img.shape[1]
                cs.height[l] = img.shape[0]
                cs.img[l] = img.ctypes.data
                if cam.edge_pyramid:
                    e = np.ascontiguousarray(cam.edge_pyramid[l], dtype=np.float64)
                    self._keep.append(e)
                    cs.edge[l] = e.ctypes.data
        self.ptr = L.po_scene_create(C.byref(cfg), n, self._cams, seed)

    @property
    def s(self) -> SceneS:
        return self.ptr.contents

    def close(self):
        if self.ptr:
            lib().po_scene_destroy(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- helpers -----------------------------------------------------
    def seed_patch(self, center, cam_idx, key: int) -> Patch:
        p = Patch()
        lib().po_patch_init_seed(self.ptr, C.byref(p), darr(center), len(cam_idx), iarr(cam_idx), key)
        return p

    def expand_patch(self, center, parent_normal, parent_cams, key: int) -> Patch:
        p = Patch()
        lib().po_patch_init_expand(self.ptr, C.byref(p), darr(center), darr(parent_normal), len(parent_cams),
                                   iarr(parent_cams), key)
        return p

    def fitness(self, p: Patch, pos) -> float:
        return float(lib().po_get_fitness(self.ptr, C.byref(p), darr(pos)))

    def set_omp(self, on: bool):
        self.ptr.contents.ompParticles = 1 if on else 0

    def set_kernel_arithmetic(self, on: bool):
        """on: fdlibm exp/sin/cos + wave64 reduction trees (what the HIP kernels compute);
        off: platform libm + the reference's sequential sums."""
        self.ptr.contents.detMath = 1 if on else 0
        self.ptr.contents.treeSum = 1 if on else 0

    def set_cost_literal(self, on: bool):
        """with kernel arithmetic on: the cost keeps the reference's statements and its sequential x-outer / y-inner sums (with the
        deterministic exp / sin / cos) -- the checker of the HIP path's PAIS_ARITH=literal mode."""
        self.ptr.contents.costLiteral = 1 if on else 0

    def set_literal_variant(self, v: int):
        """CONTROL experiments (pais_oracle.h po_scene.literalVariant, bit flags): 0 the reference's statements; 1 one rounding
        of the call perturbed; 2 y-outer accumulation; 4 contracted (fused) multiply-adds; 6 both.  Literal arithmetic only."""
        self.ptr.contents.literalVariant = int(v)
