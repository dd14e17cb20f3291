#!/bin/bash
# Round-2 closing pass, run ON the GPU box from the repo root (gpurun -- 'bash tools/r02_final.sh').
R=$PWD
OUT=$R/gpurun_out/r02f
mkdir -p $OUT
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
python -m pytest tests -x -q -m gpu 2>&1 | grep -v "MIOpen\|GridwiseOp" | tail -12 > $OUT/pytest_gpu_full.log; tail -3 $OUT/pytest_gpu_full.log
python bench.py > $OUT/bench_default.log 2>&1; grep "^{" $OUT/bench_default.log > $OUT/bench_default.json; cut -c1-260 $OUT/bench_default.json
python bench.py --no-cpu-baseline --graph off > $OUT/bench_graph_off.log 2>&1; grep "^{" $OUT/bench_graph_off.log | cut -c100-180
python bench.py --no-cpu-baseline > $OUT/bench_default_again.log 2>&1; grep "^{" $OUT/bench_default_again.log | cut -c100-180
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_final -o kt -- python $R/bench.py --steps 4 --warmup 3 --no-cpu-baseline > $OUT/kt_final_bench.log 2>&1
python $R/tools/rocprof_summary.py /tmp/kt_final $OUT/cfg2_step_kernel_trace_final.md "Round 2 final, cfg 2 train step (rocprofv3 --kernel-trace --stats of python bench.py --steps 4 --warmup 3 --no-cpu-baseline)" --steady "knn_fused_kernel<28" 2
find /tmp/kt_final -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'head -40 {} > '$OUT'/rocprofv3_kernel_stats_head.csv'
grep "^{" $OUT/kt_final_bench.log | cut -c100-180
du -sh $R/gpurun_out
