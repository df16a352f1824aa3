"""Synthetic camera rigs for tests and bench.

The reference ships no images, seeds or config file -- only the five NVM
camera lines of the "pawn" scene (``/root/reference/README.md:68-72``, copied
below as data).  Scenes are therefore synthesised: a textured solid of
revolution is ray-cast into every camera, the surface carries a band-limited
procedural albedo (sum of random plane waves evaluated at the 3-D hit point, so
all views are photo-consistent), background pixels are 0 (the reference's
foreground mask, ``patch.cpp:986`` / ``mvs.cpp:860``) and foreground
intensities are clamped to [16, 240].
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import List, Tuple

import numpy as np

from .camera import Camera, rotation_to_quaternion, quaternion_to_rotation

# README.md:68-72 -- name, focal, quaternion (w x y z), centre (x y z), radial
PAWN_NVM = [
    ("pawn0013.jpg", 614.095397949, (0.705410371683, 0.160690743319, 0.671401589359, 0.160605237544),
     (-0.556085150075, 0.0481223921551, -0.00781510757143), -0.199289312888),
    ("pawn0010.jpg", 616.175537109, (0.90353903514, 0.221746421078, 0.3576944596, 0.0806247263945),
     (-0.880841878288, 0.0327703491031, -0.684201024844), -0.209314043486),
    ("pawn0011.jpg", 612.03302002, (0.85241383667, 0.2037593266, 0.469072019941, 0.108830220502),
     (-0.71971232163, 0.0433857776889, -0.492035476323), -0.207263977174),
    ("pawn0012.jpg", 611.360473633, (0.786507583571, 0.183363764635, 0.573952646995, 0.135504187104),
     (-0.608685012281, 0.0487066227347, -0.263440114899), -0.203210786458),
    ("pawn0014.jpg", 617.585876465, (0.611485687162, 0.135944898976, 0.757586998462, 0.183482834469),
     (-0.572254659063, 0.0434025057556, 0.255716172724), -0.198563271584),
]


@dataclass
class Ellipsoid:
    z: float     # centre offset along the object's axis
    rxy: float   # radius perpendicular to the axis
    rz: float    # radius along the axis


@dataclass
class SolidOfRevolution:
    center: np.ndarray
    axis: np.ndarray
    parts: List[Ellipsoid]
    waves_k: np.ndarray    # (M, 3)
    waves_phi: np.ndarray  # (M,)
    waves_amp: np.ndarray  # (M,)

    def frame(self) -> np.ndarray:
        up = self.axis / np.linalg.norm(self.axis)
        a = np.array([1.0, 0.0, 0.0]) if abs(up[0]) < 0.9 else np.array([0.0, 1.0, 0.0])
        e1 = np.cross(up, a); e1 /= np.linalg.norm(e1)
        e2 = np.cross(up, e1)
        return np.stack([e1, e2, up])  # rows

    def intersect(self, origin: np.ndarray, dirs: np.ndarray) -> np.ndarray:
        """Nearest positive hit parameter t for rays origin + t*dirs ((N,3)); inf if none."""
        Q = self.frame()
        o = Q @ (origin - self.center)
        d = dirs @ Q.T
        best = np.full(dirs.shape[0], np.inf)
        for e in self.parts:
            s = np.array([1.0 / e.rxy, 1.0 / e.rxy, 1.0 / e.rz])
            oo = (o - np.array([0.0, 0.0, e.z])) * s
            dd = d * s
            A = np.einsum("ij,ij->i", dd, dd)
            B = 2.0 * dd @ oo
            Cc = float(oo @ oo) - 1.0
            disc = B * B - 4 * A * Cc
            ok = disc > 0
            sq = np.sqrt(np.where(ok, disc, 0.0))
            t = (-B - sq) / (2 * A)
            t = np.where(ok & (t > 1e-9), t, np.inf)
            best = np.minimum(best, t)
        return best

    def albedo(self, X: np.ndarray) -> np.ndarray:
        ph = X @ self.waves_k.T + self.waves_phi
        return 128.0 + np.sin(ph) @ self.waves_amp


def make_texture(rng: np.random.Generator, n_waves: int, lam_min: float, lam_max: float, std: float):
    dirs = rng.normal(size=(n_waves, 3))
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    lam = np.exp(rng.uniform(math.log(lam_min), math.log(lam_max), size=n_waves))
    k = dirs * (2 * math.pi / lam)[:, None]
    phi = rng.uniform(0, 2 * math.pi, size=n_waves)
    amp = np.full(n_waves, std * math.sqrt(2.0 / n_waves))
    return k, phi, amp


def render(obj: SolidOfRevolution, R: np.ndarray, C: np.ndarray, focal, pp, width: int, height: int,
           chunk_rows: int = 128) -> np.ndarray:
    """Ray-cast the object into a (height, width) uint8 image (pixel-centre sampling at integer coords)."""
    img = np.zeros((height, width), dtype=np.uint8)
    us = np.arange(width, dtype=np.float64)
    for y0 in range(0, height, chunk_rows):
        y1 = min(height, y0 + chunk_rows)
        vs = np.arange(y0, y1, dtype=np.float64)
        uu, vv = np.meshgrid(us, vs)
        dc = np.stack([(uu.ravel() - pp[0]) / focal[0], (vv.ravel() - pp[1]) / focal[1], np.ones(uu.size)], axis=1)
        dw = dc @ R  # R^T applied to camera-frame directions (rows)
        t = obj.intersect(C, dw)
        hit = np.isfinite(t)
        vals = np.zeros(uu.size)
        if hit.any():
            X = C[None, :] + t[hit, None] * dw[hit]
            vals[hit] = np.clip(np.rint(obj.albedo(X)), 16, 240)
        img[y0:y1] = vals.reshape(y1 - y0, width).astype(np.uint8)
    return img


def render_gpu(obj: SolidOfRevolution, R: np.ndarray, C: np.ndarray, focal, pp, width: int, height: int, device: int = 0,
               chunk_rows: int = 1024) -> np.ndarray:
    """The statements of render() evaluated with torch on an MI355X (float64): the 128 x 12.6 MP renders of the dome
    rig take hours in numpy on the host.  Test / bench infrastructure for producing synthetic inputs -- not the product."""
    import torch
    dev = torch.device("cuda", device)
    f64 = torch.float64
    T = lambda a: torch.as_tensor(np.asarray(a, dtype=np.float64), dtype=f64, device=dev)
    Q = T(obj.frame())
    Rt, Ct = T(R), T(C)
    o = Q @ (Ct - T(obj.center))
    wk, wphi, wamp = T(obj.waves_k), T(obj.waves_phi), T(obj.waves_amp)
    img = torch.zeros((height, width), dtype=torch.uint8, device=dev)
    us = torch.arange(width, dtype=f64, device=dev)
    for y0 in range(0, height, chunk_rows):
        y1 = min(height, y0 + chunk_rows)
        vs = torch.arange(y0, y1, dtype=f64, device=dev)
        vv, uu = torch.meshgrid(vs, us, indexing="ij")
        dc = torch.stack([(uu.reshape(-1) - pp[0]) / focal[0], (vv.reshape(-1) - pp[1]) / focal[1], torch.ones(uu.numel(), dtype=f64, device=dev)], dim=1)
        dw = dc @ Rt
        d = dw @ Q.T
        best = torch.full((dw.shape[0],), float("inf"), dtype=f64, device=dev)
        for e in obj.parts:
            s = T([1.0 / e.rxy, 1.0 / e.rxy, 1.0 / e.rz])
            oo = (o - T([0.0, 0.0, e.z])) * s
            dd = d * s
            A = (dd * dd).sum(dim=1)
            B = 2.0 * (dd @ oo)
            Cc = float(oo @ oo) - 1.0
            disc = B * B - 4 * A * Cc
            ok = disc > 0
            sq = torch.sqrt(torch.where(ok, disc, torch.zeros_like(disc)))
            t = (-B - sq) / (2 * A)
            t = torch.where(ok & (t > 1e-9), t, torch.full_like(t, float("inf")))
            best = torch.minimum(best, t)
        hit = torch.isfinite(best)
        vals = torch.zeros(uu.numel(), dtype=f64, device=dev)
        if bool(hit.any()):
            X = Ct[None, :] + best[hit, None] * dw[hit]
            alb = 128.0 + torch.sin(X @ wk.T + wphi) @ wamp
            vals[hit] = torch.clamp(torch.round(alb), 16, 240)
        img[y0:y1] = vals.reshape(y1 - y0, width).to(torch.uint8)
    return img.cpu().numpy()


@dataclass
class Scene:
    name: str
    cameras: List[Camera]
    obj: SolidOfRevolution
    seeds: List[Tuple[np.ndarray, List[int]]]   # (centre, visible camera indices)


def _look_at(C: np.ndarray, target: np.ndarray, up_hint: np.ndarray) -> np.ndarray:
    z = target - C; z /= np.linalg.norm(z)
    x = np.cross(z, up_hint); x /= np.linalg.norm(x)   # image x = right
    y = np.cross(z, x)                                  # image y = down
    return np.stack([x, y, z])


def _visible_cams(obj: SolidOfRevolution, X: np.ndarray, normal: np.ndarray, cams: List[Camera],
                  min_cos: float = 0.35) -> List[int]:
    vis = []
    for i, c in enumerate(cams):
        v = X - c.center
        dist = np.linalg.norm(v)
        d = v / dist
        if -(d @ normal) < min_cos:
            continue
        t = obj.intersect(c.center, d[None, :])[0]
        if not np.isfinite(t) or abs(t - dist) > 1e-6 * max(1.0, dist):
            continue
        # must project inside the image with a margin
        Xc = c.rotation @ X + c.translation
        u = c.focal[0] * Xc[0] / Xc[2] + c.principle_point[0]
        v2 = c.focal[1] * Xc[1] / Xc[2] + c.principle_point[1]
        if not (40 <= u < c.width - 40 and 40 <= v2 < c.height - 40):
            continue
        vis.append(i)
    return vis


def _surface_normal(obj: SolidOfRevolution, X: np.ndarray) -> np.ndarray:
    Q = obj.frame()
    xo = Q @ (X - obj.center)
    best, bn = None, None
    for e in obj.parts:
        q = (xo - np.array([0, 0, e.z])) / np.array([e.rxy, e.rxy, e.rz])
        f = abs(float(q @ q) - 1.0)
        if best is None or f < best:
            best = f
            bn = q / np.array([e.rxy, e.rxy, e.rz])
    n = Q.T @ bn
    return n / np.linalg.norm(n)


def _make_seeds(obj: SolidOfRevolution, cams: List[Camera], n_seeds: int, rng: np.random.Generator,
                min_vis: int = 3, max_vis: int = 0) -> List[Tuple[np.ndarray, List[int]]]:
    """max_vis > 0: a seed keeps at most that many cameras, the most frontal ones (an SfM track is short; the path
    tracks at most PAIS_MAX_VIS = 64 cameras per patch)."""
    seeds: List[Tuple[np.ndarray, List[int]]] = []
    tries = 0
    while len(seeds) < n_seeds and tries < n_seeds * 200:
        tries += 1
        ci = int(rng.integers(0, len(cams)))
        c = cams[ci]
        u = rng.uniform(40, c.width - 40)
        v = rng.uniform(40, c.height - 40)
        dc = np.array([(u - c.principle_point[0]) / c.focal[0], (v - c.principle_point[1]) / c.focal[1], 1.0])
        dw = c.rotation.T @ dc
        t = obj.intersect(c.center, dw[None, :])[0]
        if not np.isfinite(t):
            continue
        X = c.center + t * dw
        n = _surface_normal(obj, X)
        vis = _visible_cams(obj, X, n, cams)
        if max_vis and len(vis) > max_vis:
            cosv = []
            for i in vis:
                v = cams[i].center - X
                cosv.append(float(v @ n) / float(np.linalg.norm(v)))
            keep = sorted(sorted(range(len(vis)), key=lambda k: -cosv[k])[:max_vis])
            vis = [vis[k] for k in keep]
        if len(vis) >= min_vis:
            seeds.append((X, vis))
    return seeds


def pawn_scene(width: int = 640, height: int = 480, n_seeds: int = 200, lod_ratio: float = 0.8,
               cfg_max_lod: int = 15, tex_seed: int = 1234, seed_seed: int = 5678,
               build_edges: bool = True, tex_std: float = 34.0, tex_lam=(7.0, 40.0)) -> Scene:
    """Configs 0/1 of BASELINE.json: the 5 README cameras around a textured pawn-like solid.
    tex_std / tex_lam: grey-level standard deviation and wavelength range (pixels at the object) of the surface texture.  The
    defaults give every 31 x 31 window far more variance than MvsConfig.textureVariation, so setLOD (patch.cpp:511-610) stays at
    level 0; a faint, long-wave texture (e.g. tex_std 9, tex_lam (40, 260)) makes it climb the pyramid -- the LOW-TEXTURE variant
    of the parity tests (LOD >= 1 end to end)."""
    specs = []
    for name, f, q, C, rad in PAWN_NVM:
        R = quaternion_to_rotation(q)
        specs.append((name, f, np.asarray(q, float), np.asarray(C, float), R, rad))
    # least-squares intersection of the optical axes = where the object sits
    A = np.zeros((3, 3)); b = np.zeros(3)
    for _, _, _, C, R, _ in specs:
        d = R[2]
        P = np.eye(3) - np.outer(d, d)
        A += P; b += P @ C
    X0 = np.linalg.solve(A, b)
    up = -np.mean([R[1] for _, _, _, _, R, _ in specs], axis=0)   # image y points down
    up /= np.linalg.norm(up)
    depth = float(np.mean([np.linalg.norm(X0 - C) for _, _, _, C, _, _ in specs]))
    sc = depth / 1.13 * (min(width, height) / 480.0 if False else 1.0)
    parts = [Ellipsoid(-0.13 * sc, 0.17 * sc, 0.055 * sc), Ellipsoid(-0.02 * sc, 0.10 * sc, 0.15 * sc),
             Ellipsoid(0.13 * sc, 0.08 * sc, 0.08 * sc)]
    rng = np.random.default_rng(tex_seed)
    px = depth / 614.0 * (640.0 / width)   # world size of one pixel at the object
    k, phi, amp = make_texture(rng, 32, tex_lam[0] * px, tex_lam[1] * px, tex_std)
    obj = SolidOfRevolution(X0, up, parts, k, phi, amp)
    cams: List[Camera] = []
    s = width / 640.0
    for name, f, q, C, R, rad in specs:
        focal = np.array([f * s, f * s])
        pp = np.array([float(width >> 1), float(height >> 1)])
        img = render(obj, R, C, focal, pp, width, height)
        cams.append(Camera(focal=focal, principle_point=np.array([-1.0, -1.0]), quaternion=q, center=C,
                           image=img, name=name, radial_distortion=rad).finalize(lod_ratio, cfg_max_lod, build_edges))
    seeds = _make_seeds(obj, cams, n_seeds, np.random.default_rng(seed_seed))
    return Scene("pawn", cams, obj, seeds)


def ring_scene(n_cams: int = 32, width: int = 1920, height: int = 1080, focal: float = 2000.0,
               radius: float = 3.0, elevation_deg: float = 20.0, n_seeds: int = 400, lod_ratio: float = 0.8,
               cfg_max_lod: int = 15, tex_seed: int = 2345, seed_seed: int = 6789,
               build_edges: bool = True, device=None) -> Scene:
    """Configs 2/3: cameras on a circle looking at a textured solid at the origin.  device: render / build pyramids on that GPU."""
    X0 = np.zeros(3)
    up = np.array([0.0, 0.0, 1.0])
    parts = [Ellipsoid(-0.35, 0.55, 0.22), Ellipsoid(0.0, 0.38, 0.55), Ellipsoid(0.45, 0.30, 0.30)]
    rng = np.random.default_rng(tex_seed)
    px = radius / focal
    k, phi, amp = make_texture(rng, 32, 7 * px, 40 * px, 34.0)
    obj = SolidOfRevolution(X0, up, parts, k, phi, amp)
    cams: List[Camera] = []
    el = math.radians(elevation_deg)
    for i in range(n_cams):
        az = 2 * math.pi * i / n_cams
        C = radius * np.array([math.cos(el) * math.cos(az), math.cos(el) * math.sin(az), math.sin(el)])
        R = _look_at(C, X0, up)
        # image y must point "down": flip so that world up maps to -y
        if (R[1] @ up) > 0:
            R = np.stack([-R[0], -R[1], R[2]])
        q = rotation_to_quaternion(R)
        f2 = np.array([focal, focal])
        pp = np.array([float(width >> 1), float(height >> 1)])
        if device is None:
            img = render(obj, quaternion_to_rotation(q), C, f2, pp, width, height)
        else:
            img = render_gpu(obj, quaternion_to_rotation(q), C, f2, pp, width, height, device)
        cams.append(Camera(focal=f2, principle_point=np.array([-1.0, -1.0]), quaternion=q, center=C, image=img,
                           name="ring%04d" % i).finalize(lod_ratio, cfg_max_lod, build_edges, device=device))
    seeds = _make_seeds(obj, cams, n_seeds, np.random.default_rng(seed_seed))
    return Scene("ring", cams, obj, seeds)


def dome_scene(n_cams: int = 128, width: int = 4096, height: int = 3072, focal: float = 4500.0,
               radius: float = 4.0, n_seeds: int = 2000, lod_ratio: float = 0.8, cfg_max_lod: int = 15,
               tex_seed: int = 3456, seed_seed: int = 7890, build_edges: bool = True, device=None) -> Scene:
    """Config 4: Fibonacci hemisphere of cameras.  device: render and build the pyramids on that GPU (full size:
    128 x 4096 x 3072 takes hours in numpy); build_edges = False leaves the edge maps to the library (on the fly)."""
    X0 = np.zeros(3)
    up = np.array([0.0, 0.0, 1.0])
    parts = [Ellipsoid(-0.35, 0.75, 0.3), Ellipsoid(0.0, 0.5, 0.7), Ellipsoid(0.55, 0.4, 0.4)]
    rng = np.random.default_rng(tex_seed)
    px = radius / focal
    k, phi, amp = make_texture(rng, 32, 7 * px, 40 * px, 34.0)
    obj = SolidOfRevolution(X0, up, parts, k, phi, amp)
    cams: List[Camera] = []
    ga = math.pi * (3 - math.sqrt(5))
    for i in range(n_cams):
        zc = 0.15 + 0.8 * (i + 0.5) / n_cams
        r = math.sqrt(1 - zc * zc)
        az = ga * i
        C = radius * np.array([r * math.cos(az), r * math.sin(az), zc])
        R = _look_at(C, X0, up)
        if (R[1] @ up) > 0:
            R = np.stack([-R[0], -R[1], R[2]])
        q = rotation_to_quaternion(R)
        f2 = np.array([focal, focal])
        pp = np.array([float(width >> 1), float(height >> 1)])
        if device is None:
            img = render(obj, quaternion_to_rotation(q), C, f2, pp, width, height)
        else:
            img = render_gpu(obj, quaternion_to_rotation(q), C, f2, pp, width, height, device)
        cams.append(Camera(focal=f2, principle_point=np.array([-1.0, -1.0]), quaternion=q, center=C, image=img,
                           name="dome%04d" % i).finalize(lod_ratio, cfg_max_lod, build_edges, device=device))
    seeds = _make_seeds(obj, cams, n_seeds, np.random.default_rng(seed_seed), max_vis=32)
    return Scene("dome", cams, obj, seeds)
