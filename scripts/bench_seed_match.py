"""N4 measurement: brute-force cross-checked descriptor matching (pais_seed_match) vs the oracle's CPU loop."""
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pais_mvs_amd import seed
from oracle import po
rng = np.random.default_rng(0)
for n in (2000, 8000):
    q = rng.integers(0, 256, size=(n, 128)).astype(np.float32)
    t = rng.integers(0, 256, size=(n, 128)).astype(np.float32)
    seed.match(0, q[:64], t[:64])
    t0 = time.perf_counter(); got, _ = seed.match(0, q, t); dt = time.perf_counter() - t0
    line = "n = %d x %d x 128: GPU %.1f ms (both directions, copies included; %.1f G distance terms/s)" % (n, n, dt * 1e3, 2 * n * n * 128 / dt / 1e9)
    if n <= 2000:
        L = po.lib(); fpp = C.POINTER(C.c_float)
        w = (C.c_int * n)(); d = (C.c_float * n)()
        t0 = time.perf_counter(); L.po_seed_match(n, q.ctypes.data_as(fpp), n, t.ctypes.data_as(fpp), 128, w, d); dc = time.perf_counter() - t0
        line += "; CPU oracle (1 thread) %.0f ms; identical: %s" % (dc * 1e3, list(got) == list(w))
    print(line)
