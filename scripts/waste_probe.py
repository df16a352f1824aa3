import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pais_mvs_amd import synth
from pais_mvs_amd.config import readme_config
from pais_mvs_amd.mvs import MVS
sc = synth.pawn_scene(n_seeds=200, build_edges=False)
cfg = readme_config()
for B in [int(a) for a in sys.argv[1:]] or [512]:
    m = MVS(cfg, sc.cameras, device=0, seed=42)
    for X, vis in sc.seeds: m.add_seed(X, vis)
    m.refineSeedPatches(); m.expansionPatches(B, 0)
    st = m.stats()
    m.L.pais_mvs_debug_waste_same_dir.restype = C.c_long
    m.L.pais_mvs_debug_waste_same_dir.argtypes = [C.c_void_p]
    w = m.L.pais_mvs_debug_waste_same_dir(m.h)
    print("B=%d effective %d refined %d waste %d  of which same-parent-same-direction(after an insert) %d ; inserted %d parents %d rounds %d" % (
        B, st.candidates_effective, st.candidates_refined, st.candidates_refined - st.candidates_effective, w, st.patches_inserted, st.parents_popped, st.rounds))
    m.close()
