"""Records of the first expansion round of the full-size dome under the current PAIS_TILE* environment -> npz (diagnosis)."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, ctypes as C
from pais_mvs_amd import synth, _lib
from pais_mvs_amd.config import readme_config
from pais_mvs_amd.mvs import MVS
cfg = readme_config(patchRadius=25, distWeighting=25 / 3.0, reduceNormalRange=4.0, adaptiveGradientEnable=True)
scene = synth.dome_scene(n_seeds=400, build_edges=False, device=0)
os.environ_backup = dict(os.environ)
m = MVS(cfg, scene.cameras, device=0, seed=42)
for X, vis in scene.seeds: m.add_seed(X, vis)
m.refineSeedPatches()
m.expansion_begin()
done, cands, n = m.round_begin(1024)
from pais_mvs_amd.context import Context
ctx = Context(cfg, scene.cameras, device=0, seed=42)
ctx.set_neighbor_radius(m.neighbor_radius())
lst = [cands[i] for i in range(n)]
res = ctx.refine_batch(lst)
out = np.array([[r.fitness if r.fitness < 1e300 else -1, r.center[0], r.center[1], r.center[2], r.num_cam, r.dropped, r.pso_iterations, lst[i].num_cam] for i, r in enumerate(res)])
np.save(sys.argv[1], out)
print("saved", out.shape)
