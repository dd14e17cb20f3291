#!/bin/bash
# after the small-volume rows GEMM: full GPU suite, smoke, headline bench (with cpu_baseline), eager kernel trace
R=$PWD
OUT=$R/gpurun_out/r03f
mkdir -p $OUT
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
python -m pytest tests -q -m gpu --durations=8 2>&1 | grep -v "MIOpen\|GridwiseOp" | tail -22 > $OUT/pytest_gpu_full.log; tail -4 $OUT/pytest_gpu_full.log
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.log; cut -c1-200 $OUT/bench_default.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_final -o kt -- python $R/bench.py --steps 4 --warmup 3 --no-cpu-baseline --graph off > $OUT/kt_final_bench.log 2>&1
python $R/tools/rocprof_summary.py /tmp/kt_final $OUT/cfg2_step_kernel_trace_final.md "Round 3 final, cfg 2 train step, eager (rocprofv3 --kernel-trace --stats of python bench.py --steps 4 --warmup 3 --no-cpu-baseline --graph off)" --steady "knn_fused_kernel<28" 2
find /tmp/kt_final -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'head -40 {} > '$OUT'/rocprofv3_kernel_stats_head.csv'
