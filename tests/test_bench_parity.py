"""The benchmarked output itself, checked: bench.py's exact workload (BASELINE.json configs[1]: 5-camera pawn scene
640x480, 200 seeds, README config, PSO seed 42, R(B = 4096), to convergence; mvs/mvs.cpp:196-275) through the C ABI on
the GPU, through the oracle's drivers on the host cores of the same box, and against the hash the build container
committed (tests/golden/bench_cloud_pawn.json, made by tests/golden/make_bench_golden.py from the oracle alone)."""
import json
import os
import subprocess
import sys

import pytest

from tests import common

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "bench_cloud_pawn.json")))


def _workload():
    from pais_mvs_amd import synth
    from pais_mvs_amd.config import readme_config
    return readme_config(), synth.pawn_scene(n_seeds=GOLD["seeds"], build_edges=False)


def test_golden_file_is_the_bench_workload():
    """(CPU) the committed record names bench.py's defaults; the hash itself is re-derived on the GPU box below."""
    import bench
    sys.argv = ["bench.py"]
    a = bench.parse()
    assert (a.scene, a.seeds, a.parents_per_round, a.max_rounds) == (GOLD["scene"], GOLD["seeds"], GOLD["parents_per_round"], GOLD["max_rounds"])
    assert GOLD["patches_per_step"] == 14387 and GOLD["accepted_patches"] == 10080 and len(GOLD["cloud_sha1"]) == 40


@pytest.mark.gpu
def test_bench_workload_cloud_is_the_oracles_cloud():
    from pais_mvs_amd.mvs import MVS, patches_sha1
    cfg, scene = _workload()
    m = MVS(cfg, scene.cameras, device=0, seed=GOLD["pso_seed"])
    for X, vis in scene.seeds:
        m.add_seed(X, vis)
    m.refineSeedPatches()
    m.expansionPatches(GOLD["parents_per_round"], GOLD["max_rounds"])
    st = m.stats()
    got = [(list(p.center[:]), list(p.normalS[:]), p.cams(), p.fitness, p.correlation) for p in m.patches()]
    sha = m.cloud_sha1()
    m.close()
    # the oracle on this box's host cores (kernel arithmetic, candidates evaluated ahead of the sequential replay)
    want, calls, accepted, spec = common.oracle_reconstruct(cfg, scene, GOLD["parents_per_round"], GOLD["max_rounds"], parallel=True)
    assert st.seeds_refined + st.candidates_effective == calls == GOLD["patches_per_step"]
    assert len(got) == accepted == GOLD["accepted_patches"]
    assert st.candidates_refined - st.candidates_effective == spec == GOLD["speculative_extra_refines"]
    for i, (a, b) in enumerate(zip(got, want)):
        assert a == b, (i, a, b)          # every accepted patch, in order: same bits, same camera set
    assert sha == patches_sha1(want) == GOLD["cloud_sha1"]


@pytest.mark.gpu
def test_bench_line_carries_the_committed_hash():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0", "--no-cpu-baseline"],
                         cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    c = line["config"]
    assert c["cloud_sha1"] == GOLD["cloud_sha1"] and c["cloud_sha1_expected"] == GOLD["cloud_sha1"] and c["cloud_matches_oracle_golden"] is True
    assert c["patches_per_step"] == GOLD["patches_per_step"] and c["accepted_patches"] == GOLD["accepted_patches"]


# ---- the full-size scenes (BASELINE.json configs[2] / configs[4]), bounded to their first rounds -----------------------
# tests/golden/bench_cloud_ring_r3.json / bench_cloud_dome_r2.json were made by the ORACLE on the GPU box's host cores
# (tests/golden/make_bench_golden.py --device 0: scenes rendered on the GPU as bench.py does, edge maps for the oracle
# built next to the pyramids); `bench.py --scene ring --max-rounds 3` / `--scene dome --max-rounds 2` check themselves
# against them.  The dome's rounds see up to 44 cameras: the one-pixel instantiation of the LDS-tile kernel (K > 32).
FULL = [("ring", 3, "bench_cloud_ring_r3.json"), ("dome", 2, "bench_cloud_dome_r2.json")]


@pytest.mark.parametrize("scene,rounds,name", FULL)
def test_full_size_golden_files_name_the_bench_workloads(scene, rounds, name):
    g = json.load(open(os.path.join(ROOT, "tests", "golden", name)))
    import bench
    sys.argv = ["bench.py", "--scene", scene, "--max-rounds", str(rounds)]
    a = bench.parse()
    assert (g["scene"], g["parents_per_round"], g["max_rounds"], g["pso_seed"]) == (a.scene, a.parents_per_round, a.max_rounds, 42)
    assert g["seeds"] == max(a.seeds, 400) and len(g["cloud_sha1"]) == 40 and g["accepted_patches"] > 1000


@pytest.mark.gpu
@pytest.mark.parametrize("scene,rounds,name", FULL)
def test_full_size_bench_lines_carry_the_oracles_hash(scene, rounds, name):
    g = json.load(open(os.path.join(ROOT, "tests", "golden", name)))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--scene", scene, "--max-rounds", str(rounds), "--steps", "1",
                          "--warmup", "0", "--no-cpu-baseline"], cwd=ROOT, capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0, out.stderr[-2000:]
    c = json.loads(out.stdout.strip().splitlines()[-1])["config"]
    assert c["cloud_sha1"] == g["cloud_sha1"] and c["cloud_matches_oracle_golden"] is True, (c["cloud_sha1"], g["cloud_sha1"])
    assert c["patches_per_step"] == g["patches_per_step"] and c["accepted_patches"] == g["accepted_patches"]
    assert c["speculative_extra_refines_per_step"] == g["speculative_extra_refines"]


@pytest.mark.gpu
def test_bench_with_two_ranks_on_one_gpu_reports_one_agreed_cloud():
    """`bench.py --gpus 2` as the driver starts it (bench.py's own spawner here), with the two test hooks a one-GPU box needs:
    both ranks on device 0 and the records through host memory (RCCL refuses two ranks on one device).  Every rank hashes its
    replica of the cloud; the line says that they agree, and the cloud is the oracle's golden one."""
    env = dict(os.environ, PAIS_FORCE_DEVICE="0", PAIS_DIST_TRANSPORT="host", MASTER_PORT="29571")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"],
                         cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads([l for l in out.stdout.strip().splitlines() if l.startswith("{")][-1])
    c = line["config"]
    assert line["n_gpus"] == 2 and c["ranks_hold_the_same_cloud"] is True
    assert c["cloud_sha1"] == GOLD["cloud_sha1"] and c["cloud_matches_oracle_golden"] is True
    assert c["batches_sharded_per_step"] > 0 and c["exchange_bytes_per_step"] > 0
