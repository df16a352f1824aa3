"""On-disk formats: config.txt, NVM/NVM2, MVS_V3, PLY, PSR.  CPU only."""
import struct

import numpy as np
import pytest

# the documented config.txt values (README.md:110-207) as key/value data
README_CONFIG = {"patchRadius": 15, "reduceNormalRange": 2, "adaptiveDistanceEnable": 1, "distWeighting": 5,
                 "adaptiveDifferenceEnable": 1, "diffWeighting": 16384, "adaptiveGradientEnable": 0, "gradientWeighting": 10.0,
                 "visibleCorrelation": 0.7, "depthRangeScalar": 8, "particleNum": 15, "maxIteration": 30, "cellSize": 2,
                 "maxCellPatchNum": 3, "expansionStrategy": 0, "textureVariation": 36, "minLOD": 0, "maxLOD": 15,
                 "lodRatio": 0.8, "minCamNum": 3, "minCorrelation": 0.9, "minRegionRatio": 0.15, "maxFitness": 10.0,
                 "neighborRadiusScalar": 0.01}


def test_config_txt_parser(tmp_path):
    from pais_mvs_amd import io
    from pais_mvs_amd.config import default_config, readme_config
    p = tmp_path / "config.txt"
    lines = ["### patch optimization configuration ###", "# patch radius (patch size = 2*radius+1)", ""]
    for k, v in README_CONFIG.items():
        lines += ["# default", "%s\t\t%s" % (k, v)]
    p.write_text("\r\n".join(lines) + "\r\n")          # the reference's files are CRLF
    base = default_config(gradientWeighting=3.5)
    got = io.load_config(str(p), base)
    want = readme_config(gradientWeighting=3.5)        # gradientWeighting is NOT a key of the parser
    assert got == want
    with pytest.raises(IOError):
        io.load_config(str(tmp_path / "missing.txt"), base)


def test_nvm_reader(tmp_path):
    from pais_mvs_amd import io
    from pais_mvs_amd.synth import PAWN_NVM
    p = tmp_path / "pawn.nvm"
    lines = ["NVM_V3 ", "", "5"]
    for name, f, q, c, rad in PAWN_NVM:
        lines.append("%s\t%r %r %r %r %r %r %r %r %r 0 " % ((name, f) + tuple(q) + tuple(c) + (rad,)))
    lines += ["", "2", "0.1 -0.2 0.3 255 128 7 3 0 11 1.5 -2.5 2 12 3.25 4.0 4 13 -7 8 ",
              "1 2 3 1 2 3 2 1 5 0.5 0.25 3 6 -0.5 -0.25 ", "", "0"]
    p.write_text("\n".join(lines) + "\n")
    cams, pts = io.load_nvm(str(p))
    assert len(cams) == 5 and len(pts) == 2
    for cam, (name, f, q, c, rad) in zip(cams, PAWN_NVM):
        assert cam.file_name.decode() == name and list(cam.focal) == [f, f] and list(cam.principle_point) == [-1.0, -1.0]
        assert list(cam.quaternion) == list(q) and list(cam.center) == list(c) and cam.radial_distortion == rad
    a = pts[0]
    assert list(a.center) == [0.1, -0.2, 0.3] and list(a.rgb) == [255, 128, 7] and a.num_meas == 3
    assert list(a.cam_idx[:3]) == [0, 2, 4] and list(a.feat_idx[:3]) == [11, 12, 13]
    assert [list(a.xy[i]) for i in range(3)] == [[1.5, -2.5], [3.25, 4.0], [-7.0, 8.0]]
    # NVM2: fx fy px py after the name, no radial term (fileloader.cpp:62-110)
    p2 = tmp_path / "x.nvm2"
    p2.write_text("NVM_V3\n\n1\nimg.png 600.5 601.5 320.25 240.75 1 0 0 0 0.1 0.2 0.3\n\n0\n")
    cams2, pts2 = io.load_nvm(str(p2), nvm2=True)
    assert len(cams2) == 1 and list(cams2[0].focal) == [600.5, 601.5] and list(cams2[0].principle_point) == [320.25, 240.75]
    assert list(cams2[0].center) == [0.1, 0.2, 0.3] and cams2[0].radial_distortion == 0 and pts2 == []


def test_mvs_v3_roundtrip_and_byte_layout(tmp_path):
    from pais_mvs_amd import io
    from pais_mvs_amd.config import readme_config
    from pais_mvs_amd.synth import PAWN_NVM
    cfg = readme_config(neighborRadius=0.0123)
    cams = [io.io_camera(n, (f, f), (320, 240), q, c, rad) for n, f, q, c, rad in PAWN_NVM]
    pats = [io.io_patch((0.1 * i, -0.2, 0.3), (0.5, -1.25), list(range(3 + i % 3)), 1.5 + i, 0.95) for i in range(7)]
    path = tmp_path / "exp.mvs"
    io.write_mvs(str(path), cfg, cams, pats)
    raw = path.read_bytes()
    assert raw[:7] == b"MVS_V3\n"
    assert io._L().pais_io_sizeof_mvsconfig_disk() == 160
    blob = raw[7:167]
    # offsets of the reference's natural-alignment MvsConfig
    assert struct.unpack_from("<4i", blob, 0) == (2, 15, 31, 3)
    assert struct.unpack_from("<d", blob, 16)[0] == 36.0 and struct.unpack_from("<3i", blob, 56) == (0, 15, 3)
    assert struct.unpack_from("<d", blob, 72)[0] == 2.0 and tuple(blob[80:83]) == (1, 1, 0)
    assert struct.unpack_from("<7d", blob, 88) == (5.0, 16384.0, 10.0, 0.0123, 0.01, 0.15, 8.0)
    assert struct.unpack_from("<3i", blob, 144) == (15, 30, 0)
    assert raw[167:177] == b"CAMERAS 5\n"
    ln = struct.unpack_from("<i", raw, 177)[0]
    assert raw[181:181 + ln] == b"pawn0013.jpg"
    c2, cams2, pats2 = io.load_mvs(str(path))
    assert c2 == cfg
    assert len(cams2) == 5 and len(pats2) == 7
    for a, b in zip(cams, cams2):
        assert bytes(a) == bytes(b)
    for a, b in zip(pats, pats2):
        assert list(a.center) == list(b.center) and list(a.normalS) == list(b.normalS) and a.num_cam == b.num_cam
        assert list(a.cam_idx[:a.num_cam]) == list(b.cam_idx[:b.num_cam]) and a.fitness == b.fitness and a.correlation == b.correlation
    # MVS_V2 (no embedded config)
    v2 = tmp_path / "old.mvs"
    v2.write_bytes(b"MVS_V2\n" + raw[167:])
    c3, cams3, pats3 = io.load_mvs(str(v2))
    assert c3 is None and len(cams3) == 5 and len(pats3) == 7


def test_ply_and_psr_writers(tmp_path):
    from pais_mvs_amd import io
    cen = np.array([[0.1, -0.25, 1e-7], [123456.789, 2.0, -3.5]])
    nor = np.array([[0.0, 0.0, 1.0], [0.6, -0.8, 0.0]])
    bgr = np.array([[1, 2, 3], [250, 128, 0]], np.uint8)
    ply = tmp_path / "exp.ply"
    io.write_ply(str(ply), cen, nor, bgr)
    txt = ply.read_text().splitlines()
    assert txt[0] == "ply" and txt[1] == "format ascii 1.0" and txt[2] == "element vertex 2" and txt[12] == "end_header"
    assert txt[13] == "0.1 -0.25 1e-07 0 0 1 3 2 1"           # default ostream formatting, colours written R G B
    assert txt[14] == "123457 2 -3.5 0.6 -0.8 0 0 128 250"
    psr = tmp_path / "exp.psr"
    io.write_psr(str(psr), cen, nor)
    vals = np.frombuffer(psr.read_bytes(), dtype="<f4").reshape(2, 6)
    assert np.array_equal(vals, np.hstack([cen, nor]).astype(np.float32))
