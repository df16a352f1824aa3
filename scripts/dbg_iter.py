"""Section timers of k_pso_iter (needs the instrumented variant library built ad hoc; wall_clock64 ticks = 10 ns).
Averages over all waves of a kind, i.e. including the contention of full launches."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PAIS_LIB_PATH"] = "pais_mvs_amd/csrc/variants/libpais_dbg.so"
from pais_mvs_amd import synth
from pais_mvs_amd.config import readme_config
from pais_mvs_amd.mvs import MVS
sc = synth.pawn_scene()
m = MVS(readme_config(), sc.cameras, device=0, seed=42)
for X, vis in sc.seeds: m.add_seed(X, vis)
m.refineSeedPatches(); m.expansionPatches(4096)
out = (C.c_ulonglong * 32)()
m.L.pais_dbg_read(out)
v = list(out)
for name, b in (("one wave per evaluation", 0), ("shared evaluations (2/4 waves)", 8)):
    n = max(v[b + 3], 1); ne = max(v[b + 6], 1)
    print("%s: %d waves; per wave (us): step replay %.2f  constants -> LDS %.2f  evaluation %.2f  [of it: prologue %.2f  pixel loop %.2f over %d waves that reached the loop end]"
          % (name, v[b + 3], v[b + 0] / n / 100, v[b + 1] / n / 100, v[b + 2] / n / 100, v[b + 4] / ne / 100, v[b + 5] / ne / 100, v[b + 6]))
