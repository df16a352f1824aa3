import sys; sys.path.insert(0,'.')
from pais_mvs_amd import synth
from pais_mvs_amd.config import readme_config
from pais_mvs_amd.mvs import MVS
cfg = readme_config(patchRadius=25, distWeighting=25 / 3.0, reduceNormalRange=4.0, adaptiveGradientEnable=True)
scene = synth.dome_scene(n_seeds=400, build_edges=False, device=0)
m = MVS(cfg, scene.cameras, device=0, seed=42)
for X, vis in scene.seeds: m.add_seed(X, vis)
m.refineSeedPatches(); m.expansionPatches(1024, 3)
for l in m.round_log(): print(l.n, l.has_seeds, l.max_num_cam, round(l.refine_ms,1))
import collections
print(collections.Counter(p.num_cam for p in m.patches()))
