"""Kernel-only throughput of the cost evaluation (k_fitness): evals/s at saturation."""
import sys, os, time, ctypes as C
os.environ["PAIS_FINE_TIMING"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pais_mvs_amd import synth, _lib
from pais_mvs_amd.config import readme_config
from pais_mvs_amd.context import Context, make_candidate
sc = synth.pawn_scene(n_seeds=64, build_edges=False)
# MB_CFG='adaptiveDifferenceEnable=0,...': timing experiments with parts of the cost switched off
cfg = readme_config(**{k: bool(int(v)) for k, v in (kv.split('=') for kv in os.environ.get('MB_CFG', '').split(',') if kv)})
ctx = Context(cfg, sc.cameras, 0, 42)
# refine the seeds on the GPU to get realistic patch states
cands = []
from pais_mvs_amd.mvs import MVS
m = MVS(cfg, sc.cameras, device=0, seed=42)
for X, vis in sc.seeds: m.add_seed(X, vis)
m.refineSeedPatches()
ps = m.patches()
states = []
for p in ps:
    st = _lib.PatchState(); st.ray[:] = p.ray[:]; st.ref_cam = p.ref_cam; st.lod = p.lod; st.num_cam = p.num_cam
    for k in range(p.num_cam): st.cam_idx[k] = p.cam_idx[k]
    states.append(st)
rng = np.random.default_rng(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 400000
group = int(sys.argv[2]) if len(sys.argv) > 2 else 15   # consecutive evaluations of one patch state: the particles of a candidate
idx = np.repeat(rng.integers(0, len(states), (n + group - 1) // group), group)[:n].astype(np.int32)
parts = np.array([[ps[i].normalS[0] + rng.normal(0, .05), ps[i].normalS[1] + rng.normal(0, .05), ps[i].depth * (1 + rng.normal(0, 1e-3))] for i in idx])
K = np.mean([ps[i].num_cam for i in idx])
best = 0.0
for rep in range(3):
    t0 = time.perf_counter(); out = ctx.fitness_batch(states, idx, parts); dt = time.perf_counter() - t0
    ks = _lib.KernelStats(); ctx.L.pais_get_kernel_stats(ctx.h, C.byref(ks), 1)
    kt = ks.eval_ms / 1e3
    print("evals %d  K %.2f  call %.1f ms  kernel %.2f ms  %.1f M evals/s  (%.1f G taps/s) finite %.3f" % (n, K, dt * 1e3, kt * 1e3, n / kt / 1e6, n * 961 * K / kt / 1e9, np.mean(out < 1e300)), flush=True)
    best = max(best, n / kt)
# MB_JSON=path: the saturated rate as a record (copied to profiles/microbench_eval.json; bench.py quotes it)
if os.environ.get("MB_JSON"):
    import json
    json.dump({"evals_per_s": best, "evals": n, "mean_cameras": float(K), "scene": "pawn 640x480 (refined seeds' patch states)",
               "kernel": "k_fitness (the evaluation code of k_pso_eval2, one wave per evaluation)",
               "cmd": "python scripts/microbench_eval.py %d" % n}, open(os.environ["MB_JSON"], "w"), indent=1)
