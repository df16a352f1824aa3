"""Builds tuning variants of libpais_hip.so into pais_mvs_amd/csrc/variants/ (git-ignored .so files)."""
import os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
csrc = os.path.join(root, "pais_mvs_amd", "csrc")
out = os.path.join(csrc, "variants"); os.makedirs(out, exist_ok=True)
variants = {"g4": [], "g4w4": ["-DPAIS_EVAL_MIN_WAVES=4"], "g2": ["-DPAIS_TAP_GROUP=2"], "g2w4": ["-DPAIS_TAP_GROUP=2", "-DPAIS_EVAL_MIN_WAVES=4"],
            "g1w4": ["-DPAIS_TAP_GROUP=1", "-DPAIS_EVAL_MIN_WAVES=4"], "g2w5": ["-DPAIS_TAP_GROUP=2", "-DPAIS_EVAL_MIN_WAVES=5"],
            "g1w6": ["-DPAIS_TAP_GROUP=1", "-DPAIS_EVAL_MIN_WAVES=6"]}
for name, flags in variants.items():
    if len(sys.argv) > 1 and name not in sys.argv[1:]: continue
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-Wno-unused-result"] + flags + \
          [os.path.join(csrc, f) for f in ("pais_kernels.hip", "pais_capi.hip", "pais_mvs.hip")] + ["-o", os.path.join(out, "libpais_hip_%s.so" % name),
           "-Rpass-analysis=kernel-resource-usage"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    info = [l.strip() for l in r.stderr.splitlines() if "k_pso_eval" in l or ("VGPRs:" in l) or ("ScratchSize" in l) or "Occupancy" in l]
    # pick the lines following the k_pso_eval header
    txt = r.stderr
    i = txt.find("k_pso_eval")
    seg = txt[i:i + 1500]
    import re
    v = re.search(r"VGPRs: (\d+)", seg); sc = re.search(r"ScratchSize \[bytes/lane\]: (\d+)", seg); oc = re.search(r"Occupancy \[waves/SIMD\]: (\d+)", seg)
    print(name, flags, "VGPR", v.group(1) if v else "?", "scratch", sc.group(1) if sc else "?", "occ", oc.group(1) if oc else "?", "rc", r.returncode, flush=True)
