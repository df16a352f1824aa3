mkdir -p gpurun_out/r03e
PAIS_TILE=0 python scripts/dome_diff.py gpurun_out/r03e/d_legacy.npy 2>&1 | tail -1
for st in 40 4 13 7 39; do
PAIS_TILE_STRIP1=$st python scripts/dome_diff.py gpurun_out/r03e/d_s$st.npy 2>&1 | tail -1
python - <<PY
import numpy as np
a=np.load("gpurun_out/r03e/d_legacy.npy"); b=np.load("gpurun_out/r03e/d_s$st.npy")
d=np.where((a!=b).any(axis=1))[0]
print("strip $st: differing records:", len(d), "of", len(a), "num_cam", sorted(set(a[d,7])))
PY
done
