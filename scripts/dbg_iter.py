import os, sys, ctypes as C
sys.path.insert(0, "/root/repo")
os.environ["PAIS_LIB_PATH"] = "pais_mvs_amd/csrc/variants/libpais_dbg.so"
from pais_mvs_amd import synth
from pais_mvs_amd.config import readme_config
from pais_mvs_amd.mvs import MVS
sc = synth.pawn_scene()
m = MVS(readme_config(), sc.cameras, device=0, seed=42)
for X, vis in sc.seeds: m.add_seed(X, vis)
m.refineSeedPatches(); m.expansionPatches(4096)
out = (C.c_ulonglong * 16)()
m.L.pais_dbg_read(out)
v = list(out)
n = max(v[3], 1)
print("P=4 waves that evaluated:", v[3])
print("per wave (us): step replay %.2f  fill_eval_patch+syncs %.2f  eval total %.2f" % (v[0]/n/100, v[1]/n/100, v[2]/n/100))
ne = max(v[6], 1)
print("all evaluations that reached the pixel loop end: %d ; per eval (us): eval prologue %.2f  pixel loop %.2f" % (v[6], v[4]/ne/100, v[5]/ne/100))
