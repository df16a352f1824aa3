"""Quick tuning sweep on the GPU: parents-per-round x waves-per-candidate (one process per point)."""
import json, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pts = [tuple(map(int, a.split(":"))) for a in sys.argv[1:]] or [(512, 0)]
for B, W in pts:
    env = dict(os.environ)
    if W: env["PAIS_PSO_WAVES"] = str(W); env["PAIS_PSO_MODE"] = "fused"
    else: env.pop("PAIS_PSO_WAVES", None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                        "--parents-per-round", str(B)], env=env, capture_output=True, text=True)
    try:
        j = json.loads(r.stdout.strip().splitlines()[-1])
        c = j["config"]
        print("B=%5d W=%2d  %8.0f patches/s  step %7.1f ms  eff %6d spec+ %6d rounds %4d  k_pso %.0f ms (%.0f GB/s alg) after %.0f host %.0f" % (
            B, W, j["value"], j["ms_per_step"], c["patches_per_step"], c["speculative_extra_refines_per_step"], c["rounds_per_step"],
            j["kernel_ms"]["pso_pass"] / 2, j["roofline"]["achieved"], j["kernel_ms"]["k_after"] / 2,
            j["kernel_ms"]["host_enumerate"] + j["kernel_ms"]["host_commit"]), flush=True)
    except Exception as e:
        print("B=%d W=%d failed: %s %s" % (B, W, e, r.stderr[-500:]), flush=True)
