"""Round-2 tuning variants of the cost evaluation: same results bit for bit, different code shape.
    python scripts/variants_r02.py [name ...]      builds pais_mvs_amd/csrc/variants/libpais_<name>.so
    python scripts/variants_r02.py --run [name...] (on the GPU box) microbench + a short bench.py per variant
"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "pais_mvs_amd", "csrc")
OUT = os.path.join(SRC, "variants")
VARS = {
    "base": [],
    "ns2w4": ["-DPAIS_NS2_WAVES=4"],
    "ns2w2": ["-DPAIS_NS2_WAVES=2"],
}
FILES = ["pais_kernels.hip", "pais_capi.hip", "pais_mvs.hip", "pais_io.hip", "pais_pyramid.hip", "pais_seed.hip"]


def build(names):
    os.makedirs(OUT, exist_ok=True)
    procs = []
    for name in names:
        extra = VARS[name] if name in VARS else ["-D" + d for d in name.split(",")]
        so = os.path.join(OUT, "libpais_%s.so" % name.replace(",", "_").replace("=", ""))
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
               "-Wno-unused-result"] + extra + [os.path.join(SRC, f) for f in FILES] + ["-o", so]
        procs.append((name, so, subprocess.Popen(cmd)))
    for name, so, p in procs:
        assert p.wait() == 0, name
        print("built", so)


def run(names):
    res = {}
    for name in names:
        so = os.path.join(OUT, "libpais_%s.so" % name.replace(",", "_").replace("=", ""))
        env = dict(os.environ, PAIS_LIB_PATH=so)
        rates = []
        if not os.environ.get("NO_MICROBENCH"):
            mb = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "microbench_eval.py")], env=env, capture_output=True, text=True)
            rates = [float(l.split("M evals/s")[0].split()[-1]) for l in mb.stdout.splitlines() if "M evals/s" in l]
        b = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline"] +
                           os.environ.get("BENCH_ARGS", "--steps 4 --warmup 2").split(), env=env, capture_output=True, text=True)
        try:
            j = json.loads(b.stdout.strip().splitlines()[-1])
            res[name] = {"Mevals": max(rates) if rates else None, "patches_s": j["value"], "ms_step": j["ms_per_step"],
                         "units": j["config"]["patches_per_step"], "accepted": j["config"]["accepted_patches"]}
        except Exception as e:
            res[name] = {"Mevals": max(rates) if rates else None, "err": (b.stderr or "")[-400:]}
        print(name, res[name], flush=True)
    return res


if __name__ == "__main__":
    a = sys.argv[1:]
    if a and a[0] == "--run":
        names = a[1:] or sorted(f[8:-3] for f in os.listdir(OUT) if f.startswith("libpais_"))
        json.dump(run(names), open(os.path.join(ROOT, "gpurun_out", "variants.json"), "w"), indent=1)
    else:
        build(a or list(VARS))
