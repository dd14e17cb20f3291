#!/bin/bash
OUT=$PWD/gpurun_out/r04
mkdir -p $OUT
export MIOPEN_LOG_LEVEL=1
python -m pytest tests/test_gpu_fused.py tests/test_gpu_parity.py -x -q -k "dice or near_ties or grouped_conv or partial_count or bti" 2>&1 | tail -4
NEXTOU_DEBUG_BTI=1 python bench.py --workload cfg4 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_cfg4_graph.json 2> $OUT/bench_cfg4_graph.log
grep "BTI dbg" $OUT/bench_cfg4_graph.log | head -40
grep -v "GridwiseOp\|amdgpu.ids\|Warning\|warn\|BTI dbg" $OUT/bench_cfg4_graph.log | tail -4
python -c "import json;d=json.load(open('$OUT/bench_cfg4_graph.json'));print('cfg4', d['ms_per_step'], d['config']['step_replayed_as_hipgraph'], d['config']['graph_capture_error'])"
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_cfg2_bmm.json 2> $OUT/bench_cfg2_bmm.log
NEXTOU_GROUPED_GEMM_MIN_POINTS=0 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_cfg2_nobmm.json 2> $OUT/bench_cfg2_nobmm.log
for f in bench_cfg2_bmm bench_cfg2_nobmm; do python -c "import json;d=json.load(open('$OUT/$f.json'));print('$f', d['ms_per_step'])"; done
