#!/bin/bash
# A/B of the K2 + K7 kernel (NEXTOU_MR_GROUPED) — tests, the stage-2 Swin block in graph replay, the cfg-2 step
set -x
python -m pytest tests/test_gpu_fused.py tests/test_gpu_fused_goldens.py -q -m gpu -x -k "mr_aggregate_fused or swin_block_with or g5_blocks or g8_tiny" 2>&1 | tail -3
for v in 1 0; do
  echo "== NEXTOU_MR_GROUPED=$v"
  NEXTOU_MR_GROUPED=$v python tools/gnn_stage_profile.py --graph --cl --stages 2 2>/dev/null | tail -12
done
for v in 1 0 1 0; do
  echo "== bench NEXTOU_MR_GROUPED=$v"
  NEXTOU_MR_GROUPED=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"
done
