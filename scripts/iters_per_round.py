"""Per expansion round: candidates, PSO runs and the distribution of iterations (how many launches a round really needs)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pais_mvs_amd import synth, _lib
from pais_mvs_amd.config import readme_config
from pais_mvs_amd.mvs import MVS
sc = synth.pawn_scene(n_seeds=200)
cfg = readme_config()
m = MVS(cfg, sc.cameras, device=0, seed=42)
for X, vis in sc.seeds:
    m.add_seed(X, vis)
m.refineSeedPatches()
m.expansion_begin()
ctx = m.L.pais_mvs_ctx(m.h)
r = 0
while True:
    done, p, n = m.round_begin(4096)
    if done:
        break
    res = (_lib.PatchResult * max(n, 1))()
    if n:
        rc = m.L.pais_refine_batch(ctx, n, p, res)
        assert rc == 0
    its = np.array([res[i].pso_iterations for i in range(n) if res[i].pso_runs > 0])
    if len(its):
        print("round %2d n %5d ran %5d  iterations: max %2d  p90 %2d  median %2d  <30: %4.1f%%" % (r, n, len(its), its.max(), np.percentile(its, 90), np.median(its), 100.0 * (its < 30).mean()))
    m.round_commit(res, n)
    r += 1
m.expansion_end()
