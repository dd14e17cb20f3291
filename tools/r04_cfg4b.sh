#!/bin/bash
OUT=$PWD/gpurun_out/r04
mkdir -p $OUT
export MIOPEN_LOG_LEVEL=1
python -m pytest tests/test_gpu_fused.py tests/test_gpu_parity.py tests/test_gpu_parity2.py -x -q -k "dice or bti or near_ties or compound or ti_loss or cross_entropy" 2>&1 | tail -4
python -u tools/pool_basicconv_probe.py 2>&1 | grep -v "amdgpu.ids\|GridwiseOp" | tail -25 | tee $OUT/pool_basicconv_probe.txt
python bench.py --workload cfg4 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_cfg4_graph.json 2> $OUT/bench_cfg4_graph.log
NEXTOU_FUSED_DICE=0 python bench.py --workload cfg4 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_cfg4_graph_nodice.json 2> $OUT/bench_cfg4_graph_nodice.log
for f in bench_cfg4_graph bench_cfg4_graph_nodice; do python -c "import json;d=json.load(open('$OUT/$f.json'));print('$f', d['ms_per_step'], d['config']['step_replayed_as_hipgraph'], d['config']['graph_capture_error'], d['launch_profile_check'])"; grep -v "GridwiseOp\|amdgpu.ids\|Warning\|warn" $OUT/$f.log | tail -4; done
