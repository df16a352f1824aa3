"""`python -m pais_mvs_amd.reconstruct scene.nvm [--config config.txt] [--out DIR]`

The reference's `TMVS.exe -r` verb (TMVS.cpp:76-122) on the MI355X path: load NVM/NVM2 (+ images via PIL),
apply config.txt on top of the compiled-in defaults, refine the seeds, expand, write exp.mvs / exp.ply /
exp.psr.  Seed triangulation (`reCentering`, patch.cpp:67-112) and SIFT seeding are not part of this path
(SURVEY 8f N4): NVM points are used as they are."""
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


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("scene")
    ap.add_argument("--config", default="config.txt")
    ap.add_argument("--out", default=".")
    ap.add_argument("--parents-per-round", type=int, default=4096)
    ap.add_argument("--device", type=int, default=0)
    a = ap.parse_args(argv)
    cfg = default_config()
    if os.path.exists(a.config):
        cfg = io.load_config(a.config, cfg)
    nvm2 = a.scene.lower().endswith(".nvm2")
    cams_io, pts = io.load_nvm(a.scene, nvm2=nvm2)
    base = os.path.dirname(os.path.abspath(a.scene))
    cams = []
    for c in cams_io:
        name = c.file_name.decode()
        path = name if os.path.isabs(name) else os.path.join(base, name)
        gray, rgb = load_image_gray(path)
        cams.append(Camera(focal=np.array(c.focal[:]), principle_point=np.array(c.principle_point[:]),
                           quaternion=np.array(c.quaternion[:]), center=np.array(c.center[:]), image=gray, name=name,
                           rgb=rgb, radial_distortion=c.radial_distortion).finalize(cfg.lodRatio, cfg.maxLOD,
                                                                                    build_edges=cfg.adaptiveGradientEnable))
    m = MVS(cfg, cams, device=a.device)
    for p in pts:
        m.add_seed(p.center[:], list(p.cam_idx[:p.num_meas]))
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
