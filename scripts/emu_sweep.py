#!/usr/bin/env python
"""One-GPU emulation of rank 0 of a world of N (include/pais_mvs.h pais_mvs_emulate) under several environment settings:
    python scripts/emu_sweep.py --scene ring --B 16384 --world 8 NAME=ENV=V[,ENV=V] ...
Per setting: a fresh driver (the environment is read when it is created), one recorded single-rank run (also T(1) of that
setting), then warm-up + one timed emulated run.  Prints T(1), T_rank0(N), the ratio and the round statistics."""
import argparse, os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="ring"); ap.add_argument("--B", type=int, default=4096)
    ap.add_argument("--world", type=int, default=8); ap.add_argument("--max-rounds", type=int, default=0)
    ap.add_argument("--rank", type=int, default=0)
    ap.add_argument("specs", nargs="*")
    a = ap.parse_args()
    import torch
    import bench
    from pais_mvs_amd.mvs import MVS
    ns = argparse.Namespace(scene=a.scene, seeds=200, max_rounds=a.max_rounds)
    cfg, scene, _ = bench.build_scene(ns, 0)
    for spec in a.specs or ["default="]:
        name, _, envs = spec.partition("=")
        kv = dict(e.split("=", 1) for e in envs.split(",") if e)
        for k, v in kv.items():
            os.environ[k] = v
        m = MVS(cfg, scene.cameras, device=0, seed=42)

        def step():
            m.reset()
            for X, vis in scene.seeds:
                m.add_seed(X, vis)
            m.refineSeedPatches()
            m.expansionPatches(a.B, a.max_rounds)
            torch.cuda.synchronize()
            return m.stats()
        m.emulate(1)
        step()
        t0 = time.perf_counter(); st1 = step(); t1 = (time.perf_counter() - t0) * 1e3
        sha = m.cloud_sha1()
        if os.environ.get("EMU_TABLE"):
            import bisect
            edges = [0, 30, 68, 120, 200, 300, 500, 700, 1000, 1500, 2000, 4000, 8000, 16000, 1 << 30]
            t = [[0, 0.0, 0.0, 0.0] for _ in edges]
            for l in m.round_log():
                b = bisect.bisect_right(edges, l.n) - 1
                t[b][0] += 1; t[b][1] += l.refine_ms; t[b][2] += l.enumerate_ms; t[b][3] += l.commit_ms
            print("   candidates/round   rounds   refine ms        enumerate ms   commit ms      [single rank, unstreamed]")
            for e, r in zip(edges, t):
                if r[0]:
                    print("   >= %6d        %6d   %12.1f   %12.1f   %10.1f" % (e, r[0], r[1], r[2], r[3]))
        m.emulate(2, a.rank, a.world)
        step()
        t0 = time.perf_counter(); st = step(); tn = (time.perf_counter() - t0) * 1e3 - st.emu_replay_ms
        ok = m.cloud_sha1() == sha
        print("%-18s T1 %9.1f ms  T_rank%d(%d) %9.1f ms  x%.2f  rounds %d streamed %d sharded %d repl %d  host enum %.0f commit %.0f  gpu-wait %.0f  replay %.0f  same cloud %s"
              % (name, t1, a.rank, a.world, tn, t1 / tn, st.rounds, st.rounds_streamed, st.batches_sharded, st.batches_replicated,
                 st.host_enumerate_ms, st.host_commit_ms, st.gpu_refine_ms, st.emu_replay_ms, ok), flush=True)
        if os.environ.get("EMU_TABLE"):
            import bisect
            edges = [0, 30, 68, 120, 200, 300, 500, 700, 1000, 1500, 2000, 4000, 8000, 16000, 1 << 30]
            def table(log):
                t = [[0, 0.0, 0.0, 0.0] for _ in edges]
                for l in log:
                    b = bisect.bisect_right(edges, l.n) - 1
                    t[b][0] += 1; t[b][1] += l.refine_ms; t[b][2] += l.enumerate_ms; t[b][3] += l.commit_ms
                return t
            tn_ = table(m.round_log())
            print("   candidates/round   rounds   refine(wait) ms   enumerate ms   commit ms      [emulated rank]")
            for e, r in zip(edges, tn_):
                if r[0]:
                    print("   >= %6d        %6d   %12.1f   %12.1f   %10.1f" % (e, r[0], r[1], r[2], r[3]))
        m.emulate(0)
        m.close()
        for k in kv:
            os.environ.pop(k, None)

if __name__ == "__main__":
    main()
