"""Section timers of k_after (needs the instrumented variant lib; wall_clock64 = 100 MHz)."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PAIS_LIB_PATH"] = "pais_mvs_amd/csrc/variants/libpais_dbg.so"
from pais_mvs_amd import synth, _lib
from pais_mvs_amd.config import readme_config
from pais_mvs_amd.mvs import MVS
sc = synth.pawn_scene()
cfg = readme_config()
m = MVS(cfg, sc.cameras, device=0, seed=42)
for X, vis in sc.seeds: m.add_seed(X, vis)
m.refineSeedPatches(); m.expansionPatches(4096)
out = (C.c_ulonglong * 8)()
m.L.pais_dbg_read(out)
v = list(out)
n = max(v[4], 1)
print("candidates through k_after:", v[4])
print("per candidate (us, both removeInvisibleCamera calls summed): warp %.1f  dots %.1f  region %.1f ; whole k_after body %.1f" % (v[0] / n / 100, v[1] / n / 100, v[2] / n / 100, v[3] / n / 100))
