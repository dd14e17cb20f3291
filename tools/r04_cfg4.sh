#!/bin/bash
OUT=$PWD/gpurun_out/r04
mkdir -p $OUT
export MIOPEN_LOG_LEVEL=1
python -m pytest tests/test_gpu_fused_goldens.py -x -q -k "sliding or tiny" 2>&1 | tail -3
python bench.py --workload cfg4 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_cfg4_graph.json 2> $OUT/bench_cfg4_graph.log
python bench.py --workload cfg4 --steps 10 --warmup 3 --no-cpu-baseline --graph off > $OUT/bench_cfg4_eager.json 2> $OUT/bench_cfg4_eager.log
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_cfg2_k2fix.json 2> $OUT/bench_cfg2_k2fix.log
for f in bench_cfg4_graph bench_cfg4_eager bench_cfg2_k2fix; do python -c "import json;d=json.load(open('$OUT/$f.json'));print('$f', d['ms_per_step'], d['config']['step_replayed_as_hipgraph'], d['config']['graph_capture_error'], d['launch_profile_check'], d['roofline_step'])"; done
tail -3 $OUT/bench_cfg4_graph.log
