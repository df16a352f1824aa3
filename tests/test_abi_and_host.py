"""C ABI surface + host-side pieces that need no GPU."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(pais_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from pais_mvs_amd import _lib
    L = _lib.load()
    names = (_declared("pais_hip.h") + _declared("pais_mvs.h") + _declared("pais_io.h") + _declared("pais_pyramid.h") +
             _declared("pais_seed.h") + _declared("pais_test_hooks.h"))
    assert len(names) >= 30
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing
    nm = subprocess.check_output(["nm", "-D", "--defined-only", os.path.join(ROOT, "pais_mvs_amd", "csrc", "libpais_hip.so")], text=True)
    for n in names:
        assert re.search(r"\bT %s\b" % n, nm), n


def test_struct_sizes_match_ctypes_mirrors():
    from pais_mvs_amd import _lib
    L = _lib.load()
    assert L.pais_sizeof_config() == C.sizeof(_lib.Config)
    assert L.pais_sizeof_camera_desc() == C.sizeof(_lib.CameraDesc)
    assert L.pais_sizeof_candidate() == C.sizeof(_lib.Candidate)
    assert L.pais_sizeof_patch_result() == C.sizeof(_lib.PatchResult)


def test_no_gpu_means_loud_failure(pawn_small):
    """Without a HIP device the product refuses to work -- it never computes on the host instead."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from pais_mvs_amd.config import readme_config
    from pais_mvs_amd.context import Context
    with pytest.raises(RuntimeError):
        Context(readme_config(), pawn_small.cameras, device=0)
    from pais_mvs_amd.mvs import MVS
    m = MVS(readme_config(), pawn_small.cameras, device=-1)     # scheduler only
    m.add_seed(pawn_small.seeds[0][0], pawn_small.seeds[0][1])
    with pytest.raises(RuntimeError):
        m.refineSeedPatches()
    with pytest.raises(RuntimeError):
        m.expansionPatches(4, 1)
    m.close()
    from pais_mvs_amd.camera import build_pyramid_gpu
    with pytest.raises(RuntimeError):
        build_pyramid_gpu(pawn_small.cameras[0].pyramid[0], 0.8, 15, True, device=0)


def test_product_does_not_touch_the_oracle():
    """The product package / headers never import, include or link anything under oracle/."""
    bad = []
    for base in ("pais_mvs_amd", "include"):
        for dp, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                    txt = open(os.path.join(dp, f), errors="ignore").read()
                    if re.search(r"from\s+oracle|import\s+oracle|oracle/|pais_oracle|libpais_oracle|po_detmath", txt):
                        bad.append(os.path.join(dp, f))
    assert not bad, bad
    ldd = subprocess.check_output(["ldd", os.path.join(ROOT, "pais_mvs_amd", "csrc", "libpais_hip.so")], text=True)
    assert "oracle" not in ldd


def test_pyramid_and_camera_prep(pawn_small):
    from pais_mvs_amd.camera import resize_area, sobel_magnitude_normalised, max_lod
    cam = pawn_small.cameras[0]
    assert cam.max_lod == max_lod(cam.width, cam.height, 0.8, 15) and len(cam.pyramid) == cam.max_lod + 1
    for l, img in enumerate(cam.pyramid):
        assert img.dtype == np.uint8 and img.shape == (int(round(cam.height * 0.8 ** l)), int(round(cam.width * 0.8 ** l)))
    flat = np.full((40, 60), 77, np.uint8)
    assert (resize_area(flat, 0.8) == 77).all()
    e = sobel_magnitude_normalised(cam.pyramid[0])
    assert e.min() == 0.0 and e.max() == 1.0
    R = cam.rotation
    assert np.allclose(R @ R.T, np.eye(3), atol=1e-12) and abs(np.linalg.det(R) - 1) < 1e-12
    assert np.allclose(cam.translation, -R @ cam.center) and np.allclose(cam.optical_normal, R[2])


def test_new_entry_points_reject_bad_arguments(pawn_small):
    """Argument errors of the widened entry points are status codes with a message, never crashes (scheduler-only driver)."""
    from pais_mvs_amd.config import readme_config
    from pais_mvs_amd.mvs import MVS
    m = MVS(readme_config(), pawn_small.cameras, device=-1)
    ncam = len(pawn_small.cameras)
    with pytest.raises(RuntimeError):
        m.load_patch([0, 0, 0], [0.1, 0.2], [0, ncam], 1.0, 0.9)            # camera index out of range
    with pytest.raises(RuntimeError):
        m.add_seed_measured([0, 0, 0], [0, -1], [[1.0, 2.0], [3.0, 4.0]])    # negative camera index
    pid = m.load_patch(pawn_small.seeds[0][0], [1.0, 0.5], pawn_small.seeds[0][1], 0.5, 0.95)
    assert pid == 0 and m.num_patches() == 1
    # filters on a one-patch cloud: nothing to compare against, nothing deleted, no crash
    m.cellFiltering(); m.visibilityFiltering(); m.neighborCellFiltering(0.25)
    assert m.num_patches() == 1
    m.set_thin_front(-5)                                                     # clamped to 0
    m.close()


def test_headers_are_plain_c_and_link_from_c(tmp_path):
    """The drop-in boundary is a C ABI: every header under include/ compiles as C99 and as C++11 on its own (no HIP, no
    torch types in the signatures), and a C program linked against libpais_hip.so calls through it."""
    import glob
    import subprocess
    inc = os.path.join(ROOT, "include")
    headers = sorted(os.path.basename(h) for h in glob.glob(os.path.join(inc, "*.h")))
    assert {"pais_hip.h", "pais_mvs.h", "pais_io.h", "pais_pyramid.h"} <= set(headers)
    src = tmp_path / "abi.c"
    src.write_text("".join('#include "%s"\n' % h for h in headers) + """
#include <stdio.h>
#include <string.h>
int main(void)
{
    /* sizes the reference-side binding relies on (INTEGRATION.md) */
    if (sizeof(pais_patch_result) != (size_t)pais_sizeof_patch_result()) return 2;
    if (sizeof(pais_candidate) != (size_t)pais_sizeof_candidate()) return 3;
    if (sizeof(pais_config) != (size_t)pais_sizeof_config()) return 4;
    /* an entry point that needs no GPU: bad arguments are refused with a message */
    if (pais_mvs_create(NULL, 0, NULL, -1, 0, NULL) == 0) return 5;
    if (strlen(pais_mvs_last_error()) == 0) return 6;
    printf("ok %d %d\\n", (int)sizeof(pais_patch_result), PAIS_MAX_VIS);
    return 0;
}
""")
    lib_dir = os.path.join(ROOT, "pais_mvs_amd", "csrc")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", inc, "-fsyntax-only", str(src)], check=True)
    cpp = tmp_path / "abi.cpp"
    cpp.write_text("".join('#include "%s"\n' % h for h in headers) + "int main() { return 0; }\n")
    subprocess.run(["g++", "-std=c++11", "-Wall", "-Werror", "-I", inc, "-fsyntax-only", str(cpp)], check=True)
    exe = tmp_path / "abi"
    subprocess.run(["gcc", "-std=c99", "-I", inc, str(src), "-o", str(exe), "-L", lib_dir, "-lpais_hip", "-Wl,-rpath," + lib_dir],
                   check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.startswith("ok 1488 64"), (out.returncode, out.stdout, out.stderr)
