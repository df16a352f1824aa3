#!/usr/bin/env python
"""The CONTROL under DESIGN.md 5.3's word "chaos" (VERDICT r4 item 1c), CPU only.

bench.py's workload (pawn 640x480, 200 seeds, README config, R(4096)) is driven through the product's host scheduler
with records from the oracle in kernel arithmetic (= the HIP path's records bit for bit); the expansion candidates of
rounds 5..25 that tests/test_gpu_parity.py::test_literal_gate_on_expansion_candidates_of_the_bench_workload samples are
refined again by the oracle
  * in LITERAL arithmetic (the reference's statements, platform libm),
  * in three perturbed literal arithmetics -- the same real-number function with one rounding changed, with another
    accumulation order, with contracted multiply-adds (oracle/pais_oracle.h po_scene.literalVariant), i.e. what another
    compiler / flag / loop order would make of the REFERENCE's own source,
  * in kernel arithmetic (what the HIP path computes).
If the reference's own arithmetic noise branches as many candidates as the kernel arithmetic does, the branched
candidates are the optimiser's sensitivity, not a modelling error of the kernels; if it branches far fewer, the kernel
arithmetic adds sensitivity.  Writes tests/golden/literal_control_bench_workload.json (read by the tests and bench.py).

    python tests/golden/make_literal_control.py          (build container: ~4 min on 8 cores)
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    from pais_mvs_amd import synth
    from pais_mvs_amd.config import readme_config
    from tests import common
    t0 = time.time()
    cfg = readme_config()
    scene = synth.pawn_scene(n_seeds=200, build_edges=False)
    kept_c, _, rounds = common.cpu_workload_candidates(cfg, scene)
    t1 = time.time()
    out, _ = common.literal_control(cfg, scene, kept_c)
    out.update({"workload": "bench.py default (pawn 640x480, 200 seeds, R(4096)), expansion candidates of rounds 5..25, every n/160-th",
                "candidates": len(kept_c), "rounds": rounds, "variants": {str(k): v for k, v in common.LITERAL_VARIANTS.items()},
                "seconds": {"reconstruction": round(t1 - t0, 1), "control": round(time.time() - t1, 1)},
                "made_by": "tests/golden/make_literal_control.py"})
    with open(os.path.join(ROOT, "tests", "golden", "literal_control_bench_workload.json"), "w") as f:
        json.dump(out, f, indent=1)
        f.write("\n")
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
