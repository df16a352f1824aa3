#!/usr/bin/env python
"""Per-round host-side log of one pawn reconstruction (pais_mvs_get_round_log): candidates, refine ms -- under the current
environment.  python scripts/round_log.py [label]"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pais_mvs_amd import synth
from pais_mvs_amd.config import readme_config
from pais_mvs_amd.mvs import MVS
import time
cfg = readme_config()
scene = synth.pawn_scene(n_seeds=200, build_edges=False)
m = MVS(cfg, scene.cameras, device=0, seed=42)
def step():
    m.reset()
    for X, vis in scene.seeds:
        m.add_seed(X, vis)
    m.refineSeedPatches()
    m.expansionPatches(4096, 0)
for _ in range(2):
    step()
best = None
for _ in range(5):
    t0 = time.perf_counter(); step(); dt = time.perf_counter() - t0
    log = [(l.n, l.has_seeds, l.refine_ms) for l in m.round_log()]
    if best is None or dt < best[0]:
        best = (dt, log)
dt, log = best
thin = sum(ms for n, s, ms in log if not s and n <= 130)
med = sum(ms for n, s, ms in log if not s and 130 < n <= 600)
big = sum(ms for n, s, ms in log if not s and n > 600)
seed = sum(ms for n, s, ms in log if s)
print(json.dumps({"label": sys.argv[1] if len(sys.argv) > 1 else "", "ms": dt * 1e3, "sha": m.cloud_sha1()[:10], "seed_ms": seed, "thin_ms": thin, "thin_rounds": sum(1 for n, s, ms in log if not s and n <= 130),
                  "medium_ms": med, "medium_rounds": sum(1 for n, s, ms in log if not s and 130 < n <= 600), "large_ms": big,
                  "rounds": [(n, round(ms, 3)) for n, s, ms in log if not s]}))
m.close()
