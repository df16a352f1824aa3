"""`python -m pais_mvs_amd.reconstruct scene.nvm [--config config.txt] [--out DIR]`
`python -m pais_mvs_amd.reconstruct --filter cloud.mvs [--config config.txt] [--out DIR]`   (the `-f` verb, TMVS.cpp:124-172)

The reference's `TMVS.exe -r` verb (TMVS.cpp:76-122) on the MI355X path: load NVM/NVM2 (+ images via PIL),
apply config.txt on top of the compiled-in defaults, refine the seeds, expand, write exp.mvs / exp.ply /
exp.psr.  NVM points are re-triangulated from their measurements like MVS::loadNVM does (`reCentering`,
patch.cpp:67-112); SIFT seeding (featuremanager.cpp) is not part of this repository."""
from __future__ import annotations

import argparse
import os
import sys
import time

import numpy as np

from . import io
from .camera import Camera
from .config import default_config
from .mvs import MVS


def load_image_gray(path: str):
    from PIL import Image
    im = Image.open(path)
    rgb = np.asarray(im.convert("RGB"))
    # OpenCV's BGR2GRAY weights, fixed point as cv::cvtColor does for 8-bit (R 4899, G 9617, B 1868, shift 14)
    g = (rgb[..., 0].astype(np.int64) * 4899 + rgb[..., 1].astype(np.int64) * 9617 + rgb[..., 2].astype(np.int64) * 1868 + 8192) >> 14
    return np.clip(g, 0, 255).astype(np.uint8), rgb


def load_cameras(cams_io, base, cfg):
    cams = []
    for c in cams_io:
        name = c.file_name.decode()
        path = name if os.path.isabs(name) else os.path.join(base, name)
        gray, rgb = load_image_gray(path)
        cams.append(Camera(focal=np.array(c.focal[:]), principle_point=np.array(c.principle_point[:]),
                           quaternion=np.array(c.quaternion[:]), center=np.array(c.center[:]), image=gray, name=name,
                           rgb=rgb, radial_distortion=c.radial_distortion).finalize(cfg.lodRatio, cfg.maxLOD,
                                                                                    build_edges=cfg.adaptiveGradientEnable))
    return cams


def run_filtering(path: str, config: str, out: str, device: int = 0):
    """runFiltering (TMVS.cpp:124-172): PMVS filters 1-3, then the PCMVS neighbour filter, with the reference's
    output names."""
    file_cfg, cams_io, pats = io.load_mvs(path)
    cfg = file_cfg or default_config()
    if os.path.exists(config):
        cfg = io.load_config(config, cfg)
    cams = load_cameras(cams_io, os.path.dirname(os.path.abspath(path)), cfg)
    m = MVS(cfg, cams, device=device)
    for p in pats:
        m.load_patch(p.center[:], p.normalS[:], list(p.cam_idx[:p.num_cam]), p.fitness, p.correlation)
    print("patches: %d" % m.num_patches())
    t0 = time.perf_counter()
    m.cellFiltering()
    m.writeMVS(os.path.join(out, "PMVS_filter1.mvs")); m.writePLY(os.path.join(out, "PMVS_filter1.ply"))
    m.visibilityFiltering()
    m.writeMVS(os.path.join(out, "PMVS_filter2.mvs")); m.writePLY(os.path.join(out, "PMVS_filter2.ply"))
    m.neighborCellFiltering(0.25)
    m.writeMVS(os.path.join(out, "PMVS_filter3.mvs")); m.writePLY(os.path.join(out, "PMVS_filter3.ply"))
    ms = m.neighborPatchFiltering(0.25)
    m.writeMVS(os.path.join(out, "PCMVS_filter.mvs")); m.writePLY(os.path.join(out, "PCMVS_filter.ply"))
    print("patches after filtering: %d  time %.3f s (neighbour-count kernel %.3f ms)" % (m.num_patches(), time.perf_counter() - t0, ms))
    m.close()


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("scene")
    ap.add_argument("--filter", action="store_true", help="the -f verb: filter an .mvs cloud instead of reconstructing")
    ap.add_argument("--config", default="config.txt")
    ap.add_argument("--out", default=".")
    ap.add_argument("--parents-per-round", type=int, default=4096)
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--no-recenter", action="store_true", help="keep the NVM points as they are (skip MVS::reCentering)")
    a = ap.parse_args(argv)
    if a.filter:
        return run_filtering(a.scene, a.config, a.out, a.device)
    cfg = default_config()
    if os.path.exists(a.config):
        cfg = io.load_config(a.config, cfg)
    nvm2 = a.scene.lower().endswith(".nvm2")
    cams_io, pts = io.load_nvm(a.scene, nvm2=nvm2)
    base = os.path.dirname(os.path.abspath(a.scene))
    cams = load_cameras(cams_io, base, cfg)
    m = MVS(cfg, cams, device=a.device)
    for p in pts:
        # FileLoader::loadNvmPatch (fileloader.cpp:155-160): measurements are offsets from the image centre;
        # MVS::loadNVM then re-triangulates every point (reCentering, mvs.cpp:160-168)
        idx = list(p.cam_idx[:p.num_meas])
        meas = [[p.xy[i][0] + cams[c].width // 2, p.xy[i][1] + cams[c].height // 2] for i, c in enumerate(idx)]
        m.add_seed_measured(p.center[:], idx, meas, recenter=not a.no_recenter)
    t0 = time.perf_counter()
    m.refineSeedPatches()
    m.writeMVS(os.path.join(a.out, "seed.mvs"))
    m.expansionPatches(a.parents_per_round, 0)
    dt = time.perf_counter() - t0
    m.writeMVS(os.path.join(a.out, "exp.mvs"))
    m.writePLY(os.path.join(a.out, "exp.ply"))
    m.writePSR(os.path.join(a.out, "exp.psr"))
    st = m.stats()
    print("patches %d  refined %d  time %.3f s" % (m.num_patches(), st.seeds_refined + st.candidates_effective, dt))
    m.close()


if __name__ == "__main__":
    main()
