#!/usr/bin/env python
"""Golden hash of the benchmarked output: runs bench.py's exact workload (BASELINE.json configs[1]: 5-camera pawn scene
640x480, 200 seeds, README config, PSO seed 42, R(B = 4096), to convergence) through the ORACLE
(oracle/pais_oracle.c, kernel arithmetic, candidate-parallel replay) and records the SHA-1 of the accepted cloud
(pais_mvs_amd.mvs.patches_sha1) with the counts bench.py prints.  bench.py puts the hash of its own last step into
config.cloud_sha1; tests/test_bench_parity.py compares all three on the GPU box.

    python tests/golden/make_bench_golden.py [--scene pawn] [--max-rounds N]        (build container, ~10 min on 8 cores)
    python tests/golden/make_bench_golden.py --scene ring --max-rounds 3 --device 0   (GPU box: the full-size scenes are
    python tests/golden/make_bench_golden.py --scene dome --max-rounds 2 --device 0    rendered / pyramided on the GPU exactly as
                                                                                      bench.py does; the ORACLE then runs on the
                                                                                      box's host cores -- nothing of the HIP path
                                                                                      under test takes part in the records)
ring / dome: bench.py's workloads of those scenes (BASELINE.json configs[2] / configs[4] at full size, 400 seeds, R(4096)), bounded
to the first rounds; the oracle reads the edge maps of camera.cpp:72-77 (built next to the pyramids), the library evaluates
them on the fly.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def workload(scene_name, seeds, device=None):
    """cfg + scene exactly as bench.py's build_scene makes them (the oracle additionally gets the edge maps)"""
    from pais_mvs_amd import synth
    from pais_mvs_amd.config import readme_config
    if scene_name == "pawn":
        return readme_config(), synth.pawn_scene(n_seeds=seeds, build_edges=False)
    if device is None:
        raise SystemExit("the full-size %s scene is rendered on the GPU (as bench.py does): pass --device" % scene_name)
    if scene_name == "ring":
        return readme_config(adaptiveGradientEnable=True), synth.ring_scene(n_seeds=max(seeds, 400), build_edges=True, device=device)
    if scene_name == "dome":
        return (readme_config(patchRadius=25, distWeighting=25 / 3.0, reduceNormalRange=4.0, adaptiveGradientEnable=True),
                synth.dome_scene(n_seeds=max(seeds, 400), build_edges=True, device=device))
    raise SystemExit("unknown scene")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="pawn")
    ap.add_argument("--seeds", type=int, default=200)
    ap.add_argument("--B", type=int, default=4096)
    ap.add_argument("--max-rounds", type=int, default=0)
    ap.add_argument("--device", type=int, default=None)
    ap.add_argument("--out-dir", default=os.path.join(ROOT, "tests", "golden"))
    a = ap.parse_args()
    from pais_mvs_amd.mvs import patches_sha1
    from tests import common
    cfg, scene = workload(a.scene, a.seeds, a.device)
    t0 = time.time()
    rows, calls, accepted, spec = common.oracle_reconstruct(cfg, scene, a.B, a.max_rounds, parallel=True)
    out = {"scene": a.scene, "seeds": len(scene.seeds), "parents_per_round": a.B, "max_rounds": a.max_rounds, "pso_seed": 42,
           "patches_per_step": int(calls), "accepted_patches": int(accepted), "speculative_extra_refines": int(spec),
           "cloud_sha1": patches_sha1(rows), "oracle_seconds": round(time.time() - t0, 1),
           "made_by": "tests/golden/make_bench_golden.py (oracle, kernel arithmetic, po_mvs_set_parallel)"}
    name = "bench_cloud_%s%s.json" % (a.scene, ("_r%d" % a.max_rounds) if a.max_rounds else "")
    os.makedirs(a.out_dir, exist_ok=True)
    with open(os.path.join(a.out_dir, name), "w") as f:
        json.dump(out, f, indent=1)
        f.write("\n")
    print(json.dumps(out))


if __name__ == "__main__":
    main()
