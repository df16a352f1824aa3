import sys, os, subprocess, json, pickle
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np
    from pais_mvs_amd import synth
    from pais_mvs_amd.config import readme_config
    from pais_mvs_amd.mvs import MVS
    sc = synth.pawn_scene(n_seeds=200, build_edges=False)
    m = MVS(readme_config(), sc.cameras, device=0, seed=42)
    for X, vis in sc.seeds: m.add_seed(X, vis)
    cands, n = m.seed_begin()
    import ctypes as C
    from pais_mvs_amd import _lib
    out = (_lib.PatchResult * n)()
    _lib.check(m.L.pais_refine_batch(m.ctx_handle, n, cands, out))
    res = [(r.dropped, r.pso_runs, r.pso_iterations, r.pso_evals, r.fitness, list(r.center), r.num_cam, r.stage) for r in out]
    m.seed_commit(out, n)
    m.expansion_begin()
    for rnd in range(4):
        done, cands, n = m.round_begin(256)
        out = (_lib.PatchResult * max(n, 1))()
        _lib.check(m.L.pais_refine_batch(m.ctx_handle, n, cands, out))
        res += [("round%d" % rnd, n)] + [(r.dropped, r.pso_runs, r.pso_iterations, r.pso_evals, r.fitness, list(r.center), r.num_cam, r.stage) for r in out[:n]]
        import collections
        hist = collections.Counter((r.stage, r.dropped, r.pso_runs, min(r.pso_iterations, 99) // 5 * 5) for r in out[:n])
        print("round", rnd, "n", n, "hist(stage,dropped,runs,its/5*5):", sorted(hist.items())[:12], flush=True)
        m.round_commit(out, n)
    res.append(("patches", m.num_patches()))
    pickle.dump(res, open(sys.argv[2], "wb"))
    sys.exit(0)
outs = {}
MODES = ("iter", "split", "laststep", "persist", "fused")
for mode in MODES:
    env = dict(os.environ, PAIS_PSO_MODE=mode)
    f = "/tmp/dbg_%s.pkl" % mode
    subprocess.check_call([sys.executable, __file__, "child", f], env=env, stdout=subprocess.DEVNULL, timeout=300)
    outs[mode] = pickle.load(open(f, "rb"))
ref = outs["iter"]
for mode in MODES[1:]:
    b = outs[mode]
    diff = [i for i in range(len(ref)) if i >= len(b) or ref[i] != b[i]]
    print("%-9s records %d  differing from the default pipeline: %d" % (mode, len(b), len(diff)))
    for i in diff[:4]:
        print("   ", i, "iter", ref[i][:5], mode, b[i][:5] if i < len(b) else None)
