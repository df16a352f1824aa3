"""N2 measurement: camera pyramid + edge maps (camera.cpp:45-136) on the MI355X vs the numpy host restatement.
Algorithmic bytes per level l > 0: W*H (level-0 bytes read) + 2 * dh*W*8 (row-reduced intermediate written
and read) + dw*dh (level written); edge map of a level with n pixels: n (read) + 8n (magnitude written) + 16n (normalise)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pais_mvs_amd.camera import build_pyramid_gpu, resize_area, sobel_magnitude_normalised

for (h, w) in ((480, 640), (1080, 1920), (3072, 4096)):
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (h, w), dtype=np.uint8)
    build_pyramid_gpu(img, 0.8, 15, True, 0)  # warm-up
    t0 = time.perf_counter()
    levels, edges, ms = build_pyramid_gpu(img, 0.8, 15, True, 0)
    call = time.perf_counter() - t0
    by = 0
    for l, L in enumerate(levels):
        n = L.size
        if l > 0:
            by += w * h + 2 * L.shape[0] * w * 8 + n
        by += n + 8 * n + 16 * n
    t1 = time.perf_counter()
    if h <= 1080:
        for i in range(1, len(levels)):
            resize_area(img, 0.8 ** i)
        for L in levels:
            sobel_magnitude_normalised(L)
        host = time.perf_counter() - t1
    else:
        host = float("nan")
    print("%dx%d: %d levels, GPU kernels %.2f ms (%.0f GB/s algorithmic, %.2f of 8 TB/s), whole call incl. PCIe %.1f ms, numpy host %.0f ms"
          % (w, h, len(levels), ms, by / ms / 1e6, by / ms / 1e6 / 8000.0, call * 1e3, host * 1e3))
