"""Lane-local device arithmetic (pais_mvs_amd/csrc/pais_dev.hpp, pais_detmath.hpp) compiled for the
host by tests/host_dev_shim.cpp, checked against the oracle and glibc.  No GPU needed."""
import ctypes as C
import math
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


@pytest.fixture(scope="module")
def shim():
    bdir = os.path.join(HERE, "build")
    os.makedirs(bdir, exist_ok=True)
    so = os.path.join(bdir, "libhost_dev_shim.so")
    src = os.path.join(HERE, "host_dev_shim.cpp")
    deps = [src, os.path.join(ROOT, "pais_mvs_amd", "csrc", "pais_dev.hpp"), os.path.join(ROOT, "pais_mvs_amd", "csrc", "pais_detmath.hpp")]
    if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-o", so, src])
    S = C.CDLL(so)
    S.shim_exp_bf_mismatches.restype = C.c_long
    S.shim_exp_bf_mismatches.argtypes = [C.POINTER(C.c_double), C.c_long]
    S.shim_exp_poly_max_ulp.restype = C.c_double
    S.shim_exp_poly_max_ulp.argtypes = [C.POINTER(C.c_double), C.c_long]
    for n in ("shim_exp", "shim_exp_bf", "shim_exp_poly", "shim_sin", "shim_cos"):
        getattr(S, n).restype = C.c_double
        getattr(S, n).argtypes = [C.c_double]
    S.shim_rand31.restype = C.c_uint32
    S.shim_rand31.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32]
    S.shim_child_key.restype = C.c_uint64
    S.shim_child_key.argtypes = [C.c_uint64, C.c_int, C.c_int, C.c_int]
    S.shim_region_ratio.restype = C.c_double
    S.shim_region_ratio.argtypes = [C.c_double, C.c_double, C.c_int, C.POINTER(C.c_double)]
    dp = C.POINTER(C.c_double)
    S.shim_pso_move_all.argtypes = [C.c_int, C.c_int, C.c_double, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, dp, dp, dp, dp, dp, dp, C.c_int, dp, dp]
    return S


def _ulp(a, b):
    return 0.0 if a == b else abs(a - b) / math.ulp(b)


def test_detmath_accuracy_and_agreement(shim):
    from oracle import po
    L = po.lib()
    rng = np.random.default_rng(1)
    xs = np.concatenate([rng.uniform(-10, 10, 40000), rng.uniform(-1e-3, 1e-3, 4000), rng.uniform(-700, 700, 4000),
                         [0.0, -0.0, 1e-300, 0.5 * math.log(2), 1.5 * math.log(2), math.pi / 4, math.pi / 2, math.pi, -math.pi, 1e5]])
    me = ms = mc = 0.0
    for x in xs:
        x = float(x)
        e, s, c = shim.shim_exp(x), shim.shim_sin(x), shim.shim_cos(x)
        # the product's and the oracle's statements of the fdlibm algorithms agree bit for bit
        assert e == L.po_exp_det(x) and s == L.po_sin_det(x) and c == L.po_cos_det(x)
        if -700 < x < 700:
            me = max(me, _ulp(e, math.exp(x)))
        ms = max(ms, _ulp(s, math.sin(x))); mc = max(mc, _ulp(c, math.cos(x)))
    assert me <= 1.0 and ms <= 1.0 and mc <= 1.0, (me, ms, mc)
    assert shim.shim_exp(float("-inf")) == 0.0 and shim.shim_exp(float("inf")) == float("inf")
    assert math.isnan(shim.shim_exp(float("nan"))) and math.isnan(shim.shim_sin(float("inf")))
    assert shim.shim_exp(-800.0) == 0.0 and shim.shim_exp(710.0) == float("inf")


def test_branch_free_exp_is_bit_identical(shim):
    """det_exp_bf (what the cost kernel calls) == det_exp for every input: dense random sweeps, every fdlibm
    threshold with its neighbours, the high-word band edges, specials."""
    rng = np.random.default_rng(7)
    ln2 = math.log(2)
    parts = [rng.uniform(-60, 0, 2_000_000), rng.uniform(-1.2, 1.2, 2_000_000), rng.uniform(-800, 800, 500_000),
             -np.exp(rng.uniform(-80, 7, 500_000)), np.exp(rng.uniform(-80, 7, 200_000))]
    edges = []
    for hw in (0x3fd62e42, 0x3fd62e43, 0x3FF0A2B1, 0x3FF0A2B2, 0x3e300000, 0x3e2fffff, 0x4085E000, 0x4085DFFF, 0x40862E42,
               0x40862E41, 0x7ff00000, 0x7fefffff, 0x00100000, 0x000fffff, 0):
        for lo in (0, 1, 0x7fffffff, 0xffffffff, 0x80000000):
            for sign in (0, 1 << 63):
                edges.append(np.frombuffer(np.uint64(sign | (hw << 32) | lo).tobytes(), dtype=np.float64)[0])
    for v in (0.5 * ln2, 1.5 * ln2, 708.3964, 709.78, 745.13, 700.0):
        for s_ in (1, -1):
            e = np.float64(s_ * v)
            edges += [e, np.nextafter(e, np.inf), np.nextafter(e, -np.inf)]
    edges += [np.nan, np.inf, -np.inf, 0.0, -0.0]
    xs = np.ascontiguousarray(np.concatenate(parts + [np.array(edges, dtype=np.float64)]))
    bad = shim.shim_exp_bf_mismatches(xs.ctypes.data_as(C.POINTER(C.c_double)), len(xs))
    assert bad == 0, bad


def test_polynomial_exp_of_the_cost_weights(shim):
    """det_exp_poly (division-free exp of the adaptive weights): <= 1 ulp from the platform exp over the ranges the cost
    produces and beyond, the specials of exp, and bit for bit the oracle's statement of the same operations."""
    from oracle import po
    L = po.lib()
    rng = np.random.default_rng(5)
    for lo, hi, n in ((-6.0, 0.0, 2_000_000), (-60.0, 0.0, 1_000_000), (-700.0, 700.0, 1_000_000), (-1e-3, 1e-3, 200_000)):
        xs = np.ascontiguousarray(rng.uniform(lo, hi, n))
        assert shim.shim_exp_poly_max_ulp(xs.ctypes.data_as(C.POINTER(C.c_double)), n) <= 1.0
    for x in list(rng.uniform(-50, 1, 20000)) + [0.0, -0.0, -745.0, -745.5, -746.0, -800.0, 709.0, 709.9, 710.0, 1e5,
                                                  0.5 * math.log(2), -0.5 * math.log(2), -1e-300, 1e-300]:
        a, b = shim.shim_exp_poly(float(x)), L.po_exp_poly(float(x))
        assert a == b, (x, a, b)
    assert shim.shim_exp_poly(float("-inf")) == 0.0 and shim.shim_exp_poly(float("inf")) == float("inf")
    assert math.isnan(shim.shim_exp_poly(float("nan"))) and math.isnan(L.po_exp_poly(float("nan")))
    assert shim.shim_exp_poly(-746.0) == 0.0 and shim.shim_exp_poly(0.0) == 1.0


def test_rng_stream_identical(shim):
    from oracle import po
    L = po.lib()
    for k in (0, 1, 77, 2 ** 40 + 5):
        for r in range(3):
            for i in (0, 1, 2, 100, 4095):
                assert shim.shim_rand31(42, k, r, i) == L.po_rand31(42, k, r, i)
    for pk in (0, 1, 99999):
        for cam in range(3):
            for cx, cy in ((0, 0), (5, 7), (319, 239)):
                assert shim.shim_child_key(pk, cam, cx, cy) == L.po_child_key(pk, cam, cx, cy)


def test_region_ratio_matches_oracle_fit_ellipse(shim):
    from oracle import po
    L = po.lib()
    scn = po.SceneS()
    scn.cfg = po.config_readme()
    rng = np.random.default_rng(0)
    n = 0
    for t in range(800):
        H = np.eye(3) + rng.normal(size=(3, 3)) * np.array([[0.3, 0.3, 30], [0.3, 0.3, 30], [2e-4, 2e-4, 0.05]])
        if t % 10 == 0:
            H = np.eye(3)
        pt = rng.uniform(50, 400, size=2)
        Hc = po.darr(H.ravel())
        a = shim.shim_region_ratio(pt[0], pt[1], 15, Hc)
        b = L.po_region_ratio(C.byref(scn), po.darr(pt), Hc)
        if math.isnan(a) and math.isnan(b):
            continue
        assert abs(a - b) <= 1e-12, (a, b)
        n += 1
    assert n > 700
    assert abs(shim.shim_region_ratio(200.0, 150.0, 15, po.darr(np.eye(3).ravel())) - 1.0) < 1e-6


def test_device_pso_update_matches_oracle(shim):
    """The kernels' PSO protocol (init draws, pso_move_particle, gBest/inertia bookkeeping) emulated on the
    host reproduces po_pso_run's evaluated-position sequence exactly."""
    from oracle import po
    from tests.golden.make_golden import objective
    L = po.lib()
    dp = C.POINTER(C.c_double)

    def U(seed, key, run, k):
        return float(shim.shim_rand31(seed, key, run, k)) / 2147483647.0

    for kind in ["sphere", "plateau", "allmax"]:
        for N, maxIt in [(5, 10), (15, 30), (30, 20)]:
            key = 3
            Lo = [-2.0, -3.0, 0.0]; Up = [2.0, 1.0, 4.0]; init = [0.5, -0.5, 2.0]
            ri = [Up[d] - Lo[d] for d in range(3)]
            pos = np.zeros((N, 3)); vec = np.zeros((N, 3)); pB = np.zeros((N, 3)); nB = np.zeros((N, 3))
            for i in range(N):
                for d in range(3):
                    p = (ri[d] * U(42, key, 0, 2 * (d * N + i))) + Lo[d]
                    v = (2.0 * ri[d] * U(42, key, 0, 2 * (d * N + i) + 1)) - ri[d]
                    if i == 0:
                        p = init[d]; v = (2.0 * ri[d] * U(42, key, 0, 6 * N + d)) - ri[d]
                    pos[i, d] = p; vec[i, d] = v; pB[i, d] = p
            rec1 = []
            fit = np.zeros(N); pBF = np.zeros(N)
            for i in range(N):
                rec1.append(tuple(pos[i])); fit[i] = objective(kind, pos[i]); pBF[i] = fit[i]
            g = 0; gf = pBF[0]
            for j in range(N):
                if pBF[j] <= gf:
                    gf = pBF[j]; g = j
            iw = 0.8; it = 0
            while it < maxIt:
                disp = 0.0
                for i in range(N):
                    for d in range(3):
                        disp += abs(pos[i, d] - pB[g, d])
                disp /= (3 * N)
                conv = False
                if disp < 0.01:
                    vel = 0.0
                    for i in range(N):
                        for d in range(3):
                            vel += abs(vec[i, d])
                    conv = vel / (3 * N) < 0.01
                if conv:
                    break
                shim.shim_pso_move_all(N, min(N, 5), iw, 42, key, 0, 6 * N + 3 + 4 * (it * N), pos.ctypes.data_as(dp),
                                       vec.ctypes.data_as(dp), pB.ctypes.data_as(dp), nB.ctypes.data_as(dp),
                                       fit.ctypes.data_as(dp), pBF.ctypes.data_as(dp), g, po.darr(Lo), po.darr(Up))
                for i in range(N):
                    rec1.append(tuple(pos[i])); v = objective(kind, pos[i]); fit[i] = v
                    if v < pBF[i]:
                        pBF[i] = v; pB[i] = pos[i]
                for j in range(N):
                    if pBF[j] <= gf:
                        gf = pBF[j]; g = j
                niw = iw - 1.0 / maxIt
                iw = niw if niw > 0.4 else 0.4
                it += 1
            rec2 = []
            fn = po.FITNESS_FN(lambda p, o, rec2=rec2, kind=kind: (rec2.append((p[0], p[1], p[2])), objective(kind, [p[0], p[1], p[2]]))[1])
            r = po.RngCtx(42, key, 0, 0); res = po.PsoResult()
            L.po_pso_run(3, po.darr(Lo), po.darr(Up), C.cast(fn, C.c_void_p), None, maxIt, N, po.darr(init),
                         C.cast(L.po_rng_cb, C.c_void_p), C.addressof(r), 0, C.byref(res), None, 0, None)
            assert rec1 == rec2 and it == res.iterations and list(pB[g]) == list(res.gBest) and gf == res.gBestFitness


def test_uniform_without_division_is_the_division(tmp_path):
    """pais_dev.hpp uniform_from on the device: r * RN(1 / d) with one fused correction step instead of r / d, d = 2147483647.
    The two agree for EVERY 31-bit r (exhaustive, ~2 s on a few cores); without the correction step 9.4 M values differ."""
    import subprocess
    src = tmp_path / "divchk.c"
    src.write_text("""
#include <stdio.h>
#include <math.h>
#include <stdint.h>
int main(void) {
    const double d = 2147483647.0, y = 1.0 / 2147483647.0;
    unsigned long long bad = 0, bad0 = 0;
    #pragma omp parallel for reduction(+:bad,bad0)
    for (int64_t r = 0; r < 2147483648LL; ++r) {
        const double x = (double)r, q0 = x * y, q1 = fma(fma(-q0, d, x), y, q0), ref = x / d;
        bad += (q1 != ref);
        bad0 += (q0 != ref);
    }
    printf("%llu %llu\\n", bad, bad0);
    return 0;
}
""")
    exe = tmp_path / "divchk"
    subprocess.run(["gcc", "-O2", "-fopenmp", "-ffp-contract=off", "-o", str(exe), str(src), "-lm"], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()
    assert int(out[0]) == 0 and int(out[1]) > 0, out
