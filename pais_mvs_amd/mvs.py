"""Host mirror of PAIS::MVS for the hot path: refineSeedPatches / expansionPatches
(TMVS/mvs/mvs.h:229,231) over include/pais_mvs.h."""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence

import numpy as np

from . import _lib
from .camera import Camera
from .config import MvsConfig
from .context import camera_desc


class MvsStats(C.Structure):
    _fields_ = [("seeds_refined", C.c_int64), ("candidates_refined", C.c_int64), ("candidates_effective", C.c_int64),
                ("patches_inserted", C.c_int64), ("patches_deleted", C.c_int64), ("rounds", C.c_int64),
                ("parents_popped", C.c_int64), ("pso_evals_effective", C.c_int64),
                ("host_enumerate_ms", C.c_double), ("host_commit_ms", C.c_double), ("gpu_refine_ms", C.c_double),
                ("batches_sharded", C.c_int64), ("batches_replicated", C.c_int64), ("exchange_ms", C.c_double),
                ("exchange_bytes", C.c_int64), ("rounds_streamed", C.c_int64), ("emu_replay_ms", C.c_double),
                ("exchange_retries", C.c_int64), ("rounds_enum_sharded", C.c_int64)]


class RoundLog(C.Structure):
    _fields_ = [("n", C.c_int32), ("has_seeds", C.c_int32), ("sharded", C.c_int32), ("max_num_cam", C.c_int32),
                ("refine_ms", C.c_double), ("enumerate_ms", C.c_double), ("commit_ms", C.c_double)]


UNIQUE_ID_BYTES = 128


class UniqueId(C.Structure):
    _fields_ = [("bytes", C.c_char * UNIQUE_ID_BYTES)]


# int fn(void *user, const void *send, void *recv, size_t bytes_per_rank)
ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t)
# int fn(void *user, int n, const pais_candidate *cands, pais_patch_result *out, int has_seeds)
RECORD_SOURCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.POINTER(_lib.Candidate), C.POINTER(_lib.PatchResult), C.c_int)


def _bind(L):
    if getattr(L, "_mvs_bound", False):
        return
    vp = C.c_void_p
    L.pais_mvs_create.argtypes = [C.POINTER(_lib.Config), C.c_int, C.POINTER(_lib.CameraDesc), C.c_int, C.c_uint64,
                                  C.POINTER(vp)]
    L.pais_mvs_destroy.argtypes = [vp]
    L.pais_mvs_destroy.restype = None
    L.pais_mvs_ctx.restype = vp
    L.pais_mvs_ctx.argtypes = [vp]
    L.pais_mvs_reset.argtypes = [vp]
    L.pais_mvs_add_seed.argtypes = [vp, C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_int32)]
    L.pais_mvs_refine_seed_patches.argtypes = [vp]
    L.pais_mvs_expansion_patches.argtypes = [vp, C.c_int, C.c_int]
    L.pais_mvs_set_thin_front.argtypes = [vp, C.c_int]
    L.pais_mvs_add_seed_measured.argtypes = [vp, C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_double), C.c_int]
    L.pais_mvs_load_patch.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_int32), C.c_double, C.c_double]
    L.pais_mvs_cell_filtering.argtypes = [vp]
    L.pais_mvs_visibility_filtering.argtypes = [vp]
    L.pais_mvs_neighbor_cell_filtering.argtypes = [vp, C.c_double]
    L.pais_mvs_neighbor_patch_filtering.argtypes = [vp, C.c_double, C.POINTER(C.c_double)]
    L.pais_mvs_seed_begin.argtypes = [vp, C.POINTER(C.POINTER(_lib.Candidate)), C.POINTER(C.c_int)]
    L.pais_mvs_seed_commit.argtypes = [vp, C.POINTER(_lib.PatchResult), C.c_int]
    L.pais_mvs_expansion_begin.argtypes = [vp]
    L.pais_mvs_round_begin.argtypes = [vp, C.c_int, C.POINTER(C.POINTER(_lib.Candidate)), C.POINTER(C.c_int)]
    L.pais_mvs_round_commit.argtypes = [vp, C.POINTER(_lib.PatchResult), C.c_int]
    L.pais_mvs_expansion_end.argtypes = [vp]
    L.pais_mvs_num_patches.argtypes = [vp]
    L.pais_mvs_num_slots.argtypes = [vp]
    L.pais_mvs_get_patch.argtypes = [vp, C.c_int, C.POINTER(_lib.PatchResult), C.POINTER(C.c_int)]
    L.pais_mvs_neighbor_radius.restype = C.c_double
    L.pais_mvs_neighbor_radius.argtypes = [vp]
    L.pais_mvs_get_stats.argtypes = [vp, C.POINTER(MvsStats)]
    L.pais_mvs_get_round_log.argtypes = [vp, C.POINTER(RoundLog), C.c_int]
    L.pais_mvs_last_error.restype = C.c_char_p
    L.pais_comm_get_unique_id.argtypes = [C.POINTER(UniqueId)]
    L.pais_mvs_comm_init_rccl.argtypes = [vp, C.c_int, C.c_int, C.POINTER(UniqueId)]
    L.pais_mvs_create_ranked.argtypes = [C.POINTER(_lib.Config), C.c_int, C.POINTER(_lib.CameraDesc), C.c_int, C.c_uint64,
                                         C.c_int, C.c_int, C.POINTER(UniqueId), C.POINTER(vp)]
    L.pais_mvs_comm_init_callback.argtypes = [vp, C.c_int, C.c_int, ALLGATHER_FN, vp]
    L.pais_mvs_set_replicate_below.argtypes = [vp, C.c_int]
    L.pais_mvs_set_record_source.argtypes = [vp, RECORD_SOURCE_FN, vp]
    L.pais_mvs_emulate.argtypes = [vp, C.c_int, C.c_int, C.c_int]
    L.pais_mvs_test_inject.argtypes = [vp, C.c_int, C.c_int]
    L._mvs_bound = True


def patches_sha1(rows) -> str:
    """rows: (center[3], normalS[2], camIdx list, fitness, correlation) per patch in id order -> hex SHA-1 of the bytes
    FileWriter::writeMVS stores for them (io/filewriter.cpp:97-99)."""
    import hashlib
    import struct
    h = hashlib.sha1()
    for cen, ns, cams, fit, corr in rows:
        k = len(cams)
        h.update(struct.pack("<5di%di2d" % k, *cen, *ns, k, *cams, fit, corr))
    return h.hexdigest()


def get_unique_id() -> bytes:
    """ncclGetUniqueId through the C ABI: call on ONE rank and hand the 128 bytes to the others."""
    L = _lib.load()
    _bind(L)
    u = UniqueId()
    if L.pais_comm_get_unique_id(C.byref(u)) != 0:
        raise RuntimeError("pais_comm_get_unique_id failed: %s" % L.pais_mvs_last_error().decode())
    return bytes(bytearray(u))


class MVS:
    """Reconstruction driver.  device < 0: scheduler only (stepwise API)."""

    def __init__(self, cfg: MvsConfig, cameras: Sequence[Camera], device: int = 0, seed: int = 42):
        self.L = _lib.load()
        _bind(self.L)
        self.cfg = cfg
        self.cameras = list(cameras)
        keep: list = []
        n = len(self.cameras)
        descs = (_lib.CameraDesc * n)()
        for i, cam in enumerate(self.cameras):
            descs[i] = camera_desc(cam, bool(cfg.adaptiveGradientEnable), keep)
        c = cfg.to_c()
        h = C.c_void_p()
        self._check(self.L.pais_mvs_create(C.byref(c), n, descs, device, seed, C.byref(h)), "pais_mvs_create")
        self.h = h

    def _check(self, rc, what):
        if rc < 0:
            msg = self.L.pais_mvs_last_error().decode() or self.L.pais_last_error().decode()
            raise RuntimeError("%s failed (%d): %s" % (what, rc, msg))
        return rc

    def close(self):
        if getattr(self, "h", None):
            self.L.pais_mvs_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def ctx_handle(self):
        return self.L.pais_mvs_ctx(self.h)

    def reset(self):
        self._check(self.L.pais_mvs_reset(self.h), "pais_mvs_reset")

    def add_seed(self, center, cam_idx) -> int:
        cen = (C.c_double * 3)(*[float(v) for v in center])
        idx = (C.c_int32 * len(cam_idx))(*[int(v) for v in cam_idx])
        return self._check(self.L.pais_mvs_add_seed(self.h, cen, len(cam_idx), idx), "pais_mvs_add_seed")

    def add_seed_measured(self, center, cam_idx, img_points, recenter: bool = True) -> int:
        """NVM seed with its image measurements (pixels); recenter: MVS::reCentering first, as loadNVM does."""
        cen = (C.c_double * 3)(*[float(v) for v in center])
        idx = (C.c_int32 * len(cam_idx))(*[int(v) for v in cam_idx])
        flat = [float(v) for p in img_points for v in p]
        pts = (C.c_double * len(flat))(*flat)
        return self._check(self.L.pais_mvs_add_seed_measured(self.h, cen, len(cam_idx), idx, pts, 1 if recenter else 0),
                           "pais_mvs_add_seed_measured")

    # ---- MVS::refineSeedPatches / MVS::expansionPatches
    def refineSeedPatches(self):
        self._check(self.L.pais_mvs_refine_seed_patches(self.h), "pais_mvs_refine_seed_patches")

    # ---- FileLoader::loadMvsPatch + the `-f` filter verbs (TMVS.cpp:124-172)
    def load_patch(self, center, normalS, cam_idx, fitness, correlation) -> int:
        cen = (C.c_double * 3)(*[float(v) for v in center])
        ns = (C.c_double * 2)(*[float(v) for v in normalS])
        idx = (C.c_int32 * len(cam_idx))(*[int(v) for v in cam_idx])
        return self._check(self.L.pais_mvs_load_patch(self.h, cen, ns, len(cam_idx), idx, float(fitness), float(correlation)), "pais_mvs_load_patch")

    def cellFiltering(self):
        self._check(self.L.pais_mvs_cell_filtering(self.h), "pais_mvs_cell_filtering")

    def visibilityFiltering(self):
        self._check(self.L.pais_mvs_visibility_filtering(self.h), "pais_mvs_visibility_filtering")

    def neighborCellFiltering(self, neighbor_ratio: float = 0.25):
        self._check(self.L.pais_mvs_neighbor_cell_filtering(self.h, float(neighbor_ratio)), "pais_mvs_neighbor_cell_filtering")

    def neighborPatchFiltering(self, neighbor_ratio: float = 0.25) -> float:
        """GPU all-pairs neighbour counts; returns the kernel's milliseconds."""
        ms = C.c_double(0)
        self._check(self.L.pais_mvs_neighbor_patch_filtering(self.h, float(neighbor_ratio), C.byref(ms)), "pais_mvs_neighbor_patch_filtering")
        return ms.value

    # ---- multi-GPU (include/pais_mvs.h): one process per GPU, sharded refinement, one all-gather per batch
    def comm_init_rccl(self, rank: int, world: int, unique_id: bytes):
        u = UniqueId.from_buffer_copy(unique_id)
        self._check(self.L.pais_mvs_comm_init_rccl(self.h, rank, world, C.byref(u)), "pais_mvs_comm_init_rccl")

    def comm_init_callback(self, rank: int, world: int, all_gather):
        """all_gather(send: bytes-like view, bytes_per_rank) -> bytes of world * bytes_per_rank (rank order); host memory."""
        def cb(_user, send, recv, nbytes):
            try:
                out = all_gather((C.c_char * nbytes).from_address(send), nbytes)
                C.memmove(recv, bytes(out), nbytes * world)
                return 0
            except Exception:   # the C side reports the failure
                import traceback
                traceback.print_exc()
                return 1
        self._gather_cb = ALLGATHER_FN(cb)       # keep the trampoline alive
        self._check(self.L.pais_mvs_comm_init_callback(self.h, rank, world, self._gather_cb, None), "pais_mvs_comm_init_callback")

    def emulate(self, mode: int, rank: int = 0, world: int = 1):
        """include/pais_mvs.h pais_mvs_emulate: 1 record a single-rank run, 2 be rank `rank` of `world` on this one GPU, 0 off."""
        self._check(self.L.pais_mvs_emulate(self.h, int(mode), int(rank), int(world)), "pais_mvs_emulate")

    def set_replicate_below(self, per_rank: int):
        self._check(self.L.pais_mvs_set_replicate_below(self.h, int(per_rank)), "pais_mvs_set_replicate_below")

    def test_inject(self, what: int, count: int):
        """include/pais_test_hooks.h pais_mvs_test_inject: 1 ring-retry status, 2 failing buffer growth, 3 failing refinement."""
        self._check(self.L.pais_mvs_test_inject(self.h, int(what), int(count)), "pais_mvs_test_inject")

    def set_record_source(self, fn):
        """GPU-less drivers (device < 0) only: fn(n, cands_ptr, out_ptr, has_seeds) fills out[0:n]."""
        def cb(_user, n, cands, out, has_seeds):
            try:
                fn(n, cands, out, bool(has_seeds))
                return 0
            except Exception:
                import traceback
                traceback.print_exc()
                return 1
        self._record_cb = RECORD_SOURCE_FN(cb)
        self._check(self.L.pais_mvs_set_record_source(self.h, self._record_cb, None), "pais_mvs_set_record_source")

    def set_thin_front(self, thin_front: int):
        """Rounds with <= thin_front active parents take all remaining camera slots of each parent (0 = never)."""
        self._check(self.L.pais_mvs_set_thin_front(self.h, int(thin_front)), "pais_mvs_set_thin_front")

    def expansionPatches(self, parents_per_round: int = 1, max_rounds: int = 0):
        self._check(self.L.pais_mvs_expansion_patches(self.h, parents_per_round, max_rounds), "pais_mvs_expansion_patches")

    # ---- stepwise
    def seed_begin(self):
        p = C.POINTER(_lib.Candidate)()
        n = C.c_int(0)
        self._check(self.L.pais_mvs_seed_begin(self.h, C.byref(p), C.byref(n)), "pais_mvs_seed_begin")
        return p, n.value

    def seed_commit(self, results, n: int):
        self._check(self.L.pais_mvs_seed_commit(self.h, results, n), "pais_mvs_seed_commit")

    def expansion_begin(self):
        self._check(self.L.pais_mvs_expansion_begin(self.h), "pais_mvs_expansion_begin")

    def round_begin(self, parents_per_round: int):
        p = C.POINTER(_lib.Candidate)()
        n = C.c_int(0)
        rc = self._check(self.L.pais_mvs_round_begin(self.h, parents_per_round, C.byref(p), C.byref(n)), "pais_mvs_round_begin")
        return rc == 1, p, n.value

    def round_commit(self, results, n: int):
        self._check(self.L.pais_mvs_round_commit(self.h, results, n), "pais_mvs_round_commit")

    def expansion_end(self):
        self._check(self.L.pais_mvs_expansion_end(self.h), "pais_mvs_expansion_end")

    # ---- inspection
    def num_patches(self) -> int:
        return self.L.pais_mvs_num_patches(self.h)

    def num_slots(self) -> int:
        return self.L.pais_mvs_num_slots(self.h)

    def get_patch(self, i: int) -> Optional["_lib.PatchResult"]:
        r = _lib.PatchResult()
        e = C.c_int(0)
        if self.L.pais_mvs_get_patch(self.h, i, C.byref(r), C.byref(e)) != 0:
            return None
        return r

    def patches(self) -> List["_lib.PatchResult"]:
        out = []
        for i in range(self.num_slots()):
            p = self.get_patch(i)
            if p is not None:
                out.append(p)
        return out

    def neighbor_radius(self) -> float:
        return self.L.pais_mvs_neighbor_radius(self.h)

    def stats(self) -> MvsStats:
        s = MvsStats()
        self.L.pais_mvs_get_stats(self.h, C.byref(s))
        return s

    def round_log(self):
        """One RoundLog per GPU batch of the last reconstruction (seed batch first)."""
        n = self.L.pais_mvs_get_round_log(self.h, None, 0)
        buf = (RoundLog * max(n, 1))()
        n = self.L.pais_mvs_get_round_log(self.h, buf, n)
        return [buf[i] for i in range(n)]

    def cloud(self) -> np.ndarray:
        """(N, 6) array of patch centres and normals in id order."""
        ps = self.patches()
        return np.array([[*p.center[:], *p.normal[:]] for p in ps], dtype=np.float64).reshape(-1, 6)

    def cloud_sha1(self) -> str:
        """SHA-1 over the PATCHES payload of an MVS_V3 file (io/filewriter.cpp:97-99: per patch in id order centre[3],
        normalS[2], int K, int camIdx[K], fitness, correlation -- raw little-endian bytes): one string that pins every
        accepted patch, its order and its camera set."""
        return patches_sha1((p.center[:], p.normalS[:], p.cams(), p.fitness, p.correlation) for p in self.patches())

    # ---- writers (MVS::writeMVS / writePLY / writePSR, mvs.cpp:174-184 -> io/filewriter.cpp)
    def _io_cameras(self):
        from . import io
        return [io.io_camera(c.name or ("cam%04d" % i), c.focal, c.principle_point, c.quaternion, c.center, c.radial_distortion)
                for i, c in enumerate(self.cameras)]

    def colors_bgr(self) -> np.ndarray:
        """Patch colour as Patch::setImagePoint picks it (patch.cpp:649-652): the reference camera's pixel at
        the rounded projection of the centre (gray pipelines replicate the gray value)."""
        out = []
        for p in self.patches():
            if p.ref_cam < 0 or p.ref_cam not in p.cams():
                raise RuntimeError("patch without a reference camera: colours are defined through it (patch.cpp:649-652)")
            cam = self.cameras[p.ref_cam]
            k = p.cams().index(p.ref_cam)
            x, y = int(np.rint(p.imgPoint[k][0])), int(np.rint(p.imgPoint[k][1]))
            x = min(max(x, 0), cam.width - 1); y = min(max(y, 0), cam.height - 1)
            if cam.rgb is not None:
                out.append(cam.rgb[y, x, ::-1] if cam.rgb.shape[-1] == 3 else [cam.rgb[y, x]] * 3)
            else:
                out.append([cam.image[y, x]] * 3)
        return np.asarray(out, dtype=np.uint8).reshape(-1, 3)

    def writeMVS(self, path: str):
        from . import io
        cfg = self.cfg
        cfg.neighborRadius = self.neighbor_radius()
        pats = [io.io_patch(p.center[:], p.normalS[:], p.cams(), p.fitness, p.correlation) for p in self.patches()]
        io.write_mvs(path, cfg, self._io_cameras(), pats)

    def writePLY(self, path: str):
        from . import io
        c = self.cloud()
        io.write_ply(path, c[:, :3], c[:, 3:], self.colors_bgr())

    def writePSR(self, path: str):
        from . import io
        c = self.cloud()
        io.write_psr(path, c[:, :3], c[:, 3:])
