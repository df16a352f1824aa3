"""Sensitivity variants of the cost kernel (NOT products: each breaks the arithmetic on purpose to show what a
component costs).  Builds pais_mvs_amd/csrc/variants/libpais_<name>.so from patched temp copies of the sources."""
import os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "pais_mvs_amd", "csrc")
OUT = os.path.join(SRC, "variants")
VARS = {
    "base": [],
    # compiler scheduling strategies (no source change)
    "sched_ilp": "DEFINE:PAIS_DUMMY -mllvm -amdgpu-sched-strategy=max-ilp",
    "sched_mem": "DEFINE:PAIS_DUMMY -mllvm -amdgpu-sched-strategy=max-memory-clause",
    "sched_iter_ilp": "DEFINE:PAIS_DUMMY -mllvm -amdgpu-sched-strategy=iterative-ilp",
    "sched_aa": "DEFINE:PAIS_DUMMY -mllvm -amdgpu-use-aa-in-codegen",
    "ns4w2": "DEFINE:PAIS_NS=4 -DPAIS_ITER_WAVES=2",
    "ns3w2": "DEFINE:PAIS_NS=3 -DPAIS_ITER_WAVES=2",
    "ns1w4": "DEFINE:PAIS_NS=1 -DPAIS_ITER_WAVES=4",
    # read the group's homographies a second time from LDS (results unused): how much LDS slack is there?
    "lds2x": [('    double bx[G], by[G], nx[G], ny[G], w[G], rw[G];', '    { const double *Hx = Hbuf + 9 * ((c0 + G < 4) ? c0 + G : 0); for (int i_ = 0; i_ < 9 * G; ++i_) { double t_ = Hx[i_]; asm volatile("" ::"v"(t_)); } }\n    double bx[G], by[G], nx[G], ny[G], w[G], rw[G];')],
    # the same amount of extra VALU work instead (18 dependent-free adds per pair)
    "valu_extra": [('    double bx[G], by[G], nx[G], ny[G], w[G], rw[G];', '    { double e_ = x; for (int i_ = 0; i_ < 9 * G; ++i_) { e_ = e_ + y; asm volatile("" : "+v"(e_)); } }\n    double bx[G], by[G], nx[G], ny[G], w[G], rw[G];')],
    "nodiv": [("const double r = 1.0 / (w[0] * w[G - 1]);", "const double r = __builtin_amdgcn_rcp(w[0] * w[G - 1]);")],
    "noexp": [("if (useDiff) weight *= det_exp_bf(-(sad * sad) * invDiffW);", "if (useDiff) weight *= (1.0 - (sad * sad) * invDiffW);")],
    "waves5": [("#define PAIS_ITER_BOUNDS __launch_bounds__(64, 3)", "#define PAIS_ITER_BOUNDS __launch_bounds__(64, 5)")],
    "waves4": [("#define PAIS_ITER_BOUNDS __launch_bounds__(64, 3)", "#define PAIS_ITER_BOUNDS __launch_bounds__(64, 4)")],
}
def main(names):
    os.makedirs(OUT, exist_ok=True)
    for name in names:
        tmp = "/tmp/sens/%s/x/y/csrc" % name
        shutil.rmtree("/tmp/sens/%s" % name, ignore_errors=True)
        os.makedirs(tmp)
        for f in os.listdir(SRC):
            if f.endswith((".hip", ".hpp", ".h")):
                shutil.copy(os.path.join(SRC, f), tmp)
        shutil.copytree(os.path.join(ROOT, "include"), "/tmp/sens/%s/x/include" % name)
        p = os.path.join(tmp, "pais_kernels.hip")
        s = open(p).read()
        defs = []
        if isinstance(VARS[name], str):
            defs = ["-D" + VARS[name].split(":", 1)[1].split()[0]] + VARS[name].split(":", 1)[1].split()[1:]
        else:
            for a, b in VARS[name]:
                assert a in s, (name, a)
                s = s.replace(a, b)
        open(p, "w").write(s)
        so = os.path.join(OUT, "libpais_%s.so" % name)
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
                               "-Wno-unused-result"] + defs + [os.path.join(tmp, f) for f in ("pais_kernels.hip", "pais_capi.hip", "pais_mvs.hip", "pais_io.hip", "pais_pyramid.hip")] + ["-o", so])
        print("built", so)
if __name__ == "__main__":
    main(sys.argv[1:] or list(VARS))
