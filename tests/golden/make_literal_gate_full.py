#!/usr/bin/env python
"""The literal-arithmetic gate at FULL size for the ring and the dome, run ON THE GPU BOX.

bench.py's workload of the scene (BASELINE.json configs[2] / configs[4] at full size, 400 seeds, R(4096)) is driven round by
round through the stepwise entry points of the C ABI on the GPU; of rounds >= 2 every n/`--per-round`-th expansion candidate is
kept with the HIP path's record.  The kept candidates are then refined again by the ORACLE on the box's host cores in LITERAL
arithmetic (the reference's statements), in the like-for-like control variant (literal with y-outer sums and fused
multiply-adds: what another loop order and compiler make of the reference's own source) and in kernel arithmetic; the HIP
records are asserted to BE the kernel-arithmetic patches bit for bit, and the trajectories are compared per candidate exactly
as tests/test_gpu_parity.py does for the pawn workload.  Writes gpurun_out/literal_gate_<scene>_full.json (committed under
tests/golden/ once measured; bench.py --scene <scene> prints it as config.literal_gate).

    python tests/golden/make_literal_gate_full.py --scene ring --rounds 6 --device 0
    python tests/golden/make_literal_gate_full.py --scene dome --rounds 4 --device 0
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", required=True, choices=["ring", "dome"])
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--first-round", type=int, default=2)
    ap.add_argument("--per-round", type=int, default=100)
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--B", type=int, default=4096)
    a = ap.parse_args()
    from pais_mvs_amd import _lib
    from pais_mvs_amd.mvs import MVS
    from tests import common
    from tests.golden.make_bench_golden import workload
    from tests.test_oracle_modes import mode_statistics
    t0 = time.time()
    cfg, scene = workload(a.scene, 400, a.device)      # (with the edge maps the oracle reads)
    t_scene = time.time() - t0
    m = MVS(cfg, scene.cameras, device=a.device, seed=42)
    for X, vis in scene.seeds:
        m.add_seed(X, vis)
    m.refineSeedPatches()
    m.expansion_begin()
    radius = m.neighbor_radius()                       # (what setDepthRange reads during the rounds: mvs.cpp:116-141 at expansion start)
    L = m.L
    L.pais_refine_batch.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    kept_c, kept_r, sizes = [], [], []
    rnd = 0
    while rnd < a.rounds:
        done, cands, n = m.round_begin(a.B)
        if done:
            break
        out = (_lib.PatchResult * max(n, 1))()
        if n:
            assert L.pais_refine_batch(m.ctx_handle, n, cands, out) == 0
            if rnd >= a.first_round:
                for i in range(0, n, max(1, n // a.per_round)):
                    kept_c.append(common.copy_struct(cands[i]))
                    kept_r.append(common.copy_struct(out[i]))
        sizes.append(n)
        m.round_commit(out, n)
        rnd += 1
    m.expansion_end()
    m.close()
    t_gpu = time.time() - t0 - t_scene
    S = common.oracle_scene(cfg, scene)
    S.ptr.contents.cfg.neighborRadius = radius
    runs = {}
    for name, ka, var in (("literal", False, 0), ("control_6", False, 6), ("kernel", True, 0)):
        S.set_kernel_arithmetic(ka)
        S.set_literal_variant(var)
        runs[name] = common.oracle_refine_many(S, kept_c, False)
    S.set_literal_variant(0)
    st = mode_statistics(runs["literal"], runs["kernel"], hip=kept_r)   # (asserts HIP == kernel arithmetic bit for bit)
    ker = common.trajectory_split(runs["literal"], runs["kernel"])
    ctl = common.trajectory_split(runs["literal"], runs["control_6"])
    by_k = {}
    for x, y in zip(runs["literal"], runs["kernel"]):
        if x.drop or y.drop:
            continue
        t = by_k.setdefault(int(x.numCam), [0, 0])
        t[0] += 1
        t[1] += int(not (x.psoSig == y.psoSig and x.psoRuns == y.psoRuns and x.psoIters == y.psoIters))
    rep = {"scene": a.scene, "rounds": rnd, "round_sizes": sizes, "first_round_sampled": a.first_round, "candidates": ker["n"],
           "hip_equals_kernel_arithmetic_bit_for_bit": True,
           "branched": ker["branched"], "branched_fraction": ker["branched_fraction"],
           "same_trajectory_centre_max": ker["same_centre_max"], "same_trajectory_normal_max": ker["same_normal_max"],
           "branched_centre_max": ker["branched_centre_max"], "branched_normal_max": ker["branched_normal_max"],
           "beyond_1e-4": ker["beyond_1e-4"],
           "set_mismatch_on_the_same_trajectory": ker["set_mismatch_on_the_same_trajectory"],
           "set_mismatch_among_branched": ker["set_mismatch_among_branched"],
           "control_literal_y_outer_and_fused": {k: ctl[k] for k in ("branched", "branched_fraction", "beyond_1e-4", "set_mismatch_among_branched",
                                                                       "same_centre_max", "same_normal_max")},
           "by_num_cam": {str(k): {"n": v[0], "branched": v[1]} for k, v in sorted(by_k.items())},
           "seconds": {"scene": round(t_scene, 1), "gpu_rounds": round(t_gpu, 1), "oracle": round(time.time() - t0 - t_scene - t_gpu, 1)},
           "workload": "bench.py --scene %s (full size, 400 seeds, R(%d)): every ~n/%d-th expansion candidate of rounds %d..%d"
                       % (a.scene, a.B, a.per_round, a.first_round, rnd - 1),
           "made_by": "tests/golden/make_literal_gate_full.py on the GPU box (HIP records; oracle on the host cores)"}
    assert st["n"] == ker["n"]
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "literal_gate_%s_full.json" % a.scene), "w") as f:
        json.dump(rep, f, indent=1)
        f.write("\n")
    print(json.dumps(rep))


if __name__ == "__main__":
    main()
