#!/bin/bash
OUT=$PWD/gpurun_out/r04
mkdir -p $OUT
export MIOPEN_LOG_LEVEL=1
python -m pytest tests/test_gpu_fused.py tests/test_gpu_parity.py tests/test_gpu_parity2.py -x -q -k "dice or near_ties or grouped_conv or partial_count or bti or compound or ti_loss" 2>&1 | tail -4
python bench.py --workload cfg4 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_cfg4_graph.json 2> $OUT/bench_cfg4_graph.log
grep -v "GridwiseOp\|amdgpu.ids\|Warning\|warn" $OUT/bench_cfg4_graph.log | tail -4
NEXTOU_FUSED_DICE=0 python bench.py --workload cfg4 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_cfg4_graph_nodice.json 2> $OUT/bench_cfg4_graph_nodice.log
python bench.py --workload cfg4 --steps 10 --warmup 3 --no-cpu-baseline --graph off > $OUT/bench_cfg4_eager.json 2> $OUT/bench_cfg4_eager.log
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_cfg2_mid.json 2> $OUT/bench_cfg2_mid.log
for f in bench_cfg4_graph bench_cfg4_graph_nodice bench_cfg4_eager bench_cfg2_mid; do python -c "import json;d=json.load(open('$OUT/$f.json'));print('$f', d['ms_per_step'], d['config']['step_replayed_as_hipgraph'], d['config']['graph_capture_error'])"; done
