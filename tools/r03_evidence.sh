#!/bin/bash
# Round-3 evidence run (VERDICT r2 "close the evidence gaps"): K5 roofline, SQ counters of the pooled K2 forward, the bench
# line with the CPU thread sweep, the averaged step under hipGraph capture, the cfg-4 per-kernel breakdown.
# usage on the GPU box from the repo root: bash tools/r03_evidence.sh
OUT=$PWD/gpurun_out/r03
mkdir -p $OUT
export TMPDIR=/tmp
python tools/kernel_bench.py --bti --iters 10 --json $OUT/kernel_bench_k5.json > $OUT/kernel_bench_k5.txt 2>&1; tail -25 $OUT/kernel_bench_k5.txt
python tools/kernel_bench.py --cfg 2 --iters 10 > $OUT/kernel_bench_cfg2_start.txt 2>&1
bash tools/pmc_sq.sh "s3 Pool" "" r03_s3_pool > $OUT/pmc_sq_s3pool.log 2>&1
python bench.py --steps 10 --warmup 3 --cpu-thread-sweep > $OUT/bench_start.json 2> $OUT/bench_start.log; cat $OUT/bench_start.json
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --force-averager --graph on > $OUT/bench_start_averager_graph.json 2> $OUT/bench_start_averager_graph.log; tail -2 $OUT/bench_start_averager_graph.log; cat $OUT/bench_start_averager_graph.json
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --force-averager --graph off > $OUT/bench_start_averager_eager.json 2> $OUT/bench_start_averager_eager.log; cat $OUT/bench_start_averager_eager.json
python bench.py --steps 6 --warmup 3 --no-cpu-baseline --workload cfg4 > $OUT/bench_cfg4_start.json 2> $OUT/bench_cfg4_start.log; cat $OUT/bench_cfg4_start.json
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/kt_cfg4 -o kt -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 3 --no-cpu-baseline --workload cfg4 --graph off > $OUT/kt_cfg4.log 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/kt_cfg2 -o kt -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 3 --no-cpu-baseline --graph off > $OUT/kt_cfg2.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocprof_summary.py $OUT/kt_cfg4 $OUT/cfg4_step_kernel_trace.md "cfg 4 step (Dice + CE + BTI), eager" --steady "knn_fused_kernel<28" 2 || true
python tools/rocprof_summary.py $OUT/kt_cfg2 $OUT/cfg2_step_kernel_trace_start.md "cfg 2 step, eager, round-3 start" --steady "knn_fused_kernel<28" 2 || true
rm -rf $OUT/kt_cfg4/*.db $OUT/kt_cfg2/*.db 2>/dev/null; find $OUT -name "*.db" -size +20M -delete
ls -la $OUT
