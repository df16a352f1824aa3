#!/bin/bash
# round 6: the two-level canonical sums (K >= 13) and the symmetric split kernel: parity, A/B, dome goldens regenerated on the box
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "dome or ring_all_weights or literal_arithmetic_cost or many_cameras" > gpurun_out/t7.log 2>&1; grep -E "passed|failed" gpurun_out/t7.log | tail -2
bash scripts/dome_env_ab.sh ab_sym 10 2 split=PAIS_TILE_SPLIT=1 old=PAIS_TILE_SPLIT=0
timeout 900 python tests/golden/make_bench_golden.py --scene dome --max-rounds 2 --device 0 > gpurun_out/golden_dome.log 2>&1; cp tests/golden/bench_cloud_dome_r2.json gpurun_out/bench_cloud_dome_r2.json; tail -2 gpurun_out/golden_dome.log
PAIS_TILE_SPLIT=1 timeout 900 python tests/golden/make_literal_gate_full.py --scene dome --rounds 8 --per-round 120 --device 0 > gpurun_out/gate_dome.log 2>&1; tail -2 gpurun_out/gate_dome.log
