#!/bin/bash
# round 6: rebalanced two-level split: parity, A/B, goldens of the dome and the ring regenerated on the box
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "dome_radius25 or ring_all_weights" > gpurun_out/t10.log 2>&1; grep -E "passed|failed" gpurun_out/t10.log | tail -1
bash scripts/dome_env_ab.sh ab_bal 10 2 split=PAIS_TILE_SPLIT=1 old=PAIS_TILE_SPLIT=0
timeout 900 python tests/golden/make_bench_golden.py --scene dome --max-rounds 2 --device 0 > gpurun_out/golden_dome.log 2>&1; cp tests/golden/bench_cloud_dome_r2.json gpurun_out/bench_cloud_dome_r2.json; tail -1 gpurun_out/golden_dome.log | cut -c1-300
timeout 900 python tests/golden/make_literal_gate_full.py --scene dome --rounds 8 --per-round 120 --device 0 > gpurun_out/gate_dome.log 2>&1; tail -1 gpurun_out/gate_dome.log | cut -c1-400
timeout 900 python tests/golden/make_bench_golden.py --scene ring --max-rounds 3 --device 0 > gpurun_out/golden_ring.log 2>&1; cp tests/golden/bench_cloud_ring_r3.json gpurun_out/bench_cloud_ring_r3.json; tail -1 gpurun_out/golden_ring.log | cut -c1-300
timeout 900 python tests/golden/make_literal_gate_full.py --scene ring --rounds 12 --per-round 150 --device 0 > gpurun_out/gate_ring.log 2>&1; tail -1 gpurun_out/gate_ring.log | cut -c1-400
