for st in 6 8 3 41; do
  echo "== NS1 strip $st"; PAIS_TILE_FORCE_NS1=1 PAIS_TILE_STRIP1=$st timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "dome_radius25_many_cameras and tile" 2>&1 | tail -4
done
for st in 8 4 42; do
  echo "== NS2 strip $st"; PAIS_TILE_STRIP2=$st timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "dome_radius25_many_cameras and tile" 2>&1 | tail -4
done
