#!/bin/bash
# Round-3 closing pass, run ON the GPU box from the repo root (gpurun -- 'bash tools/r03_final.sh').
R=$PWD
OUT=$R/gpurun_out/r03f
mkdir -p $OUT
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
python -m pytest tests -q -m gpu --durations=8 2>&1 | grep -v "MIOpen\|GridwiseOp" | tail -22 > $OUT/pytest_gpu_full.log; tail -4 $OUT/pytest_gpu_full.log
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.log; cut -c1-200 $OUT/bench_default.json
NEXTOU_PW_FUSE=0 python bench.py --no-cpu-baseline > $OUT/bench_fuse0.json 2> $OUT/bench_fuse0.log; python -c "import json;print('PW_FUSE=0', json.load(open('$OUT/bench_fuse0.json'))['ms_per_step'])"
python bench.py --no-cpu-baseline > $OUT/bench_default_again.json 2> $OUT/bench_default_again.log; python -c "import json;print('default again', json.load(open('$OUT/bench_default_again.json'))['ms_per_step'])"
python bench.py --no-cpu-baseline --steps 10 --warmup 3 --workload cfg4 > $OUT/bench_cfg4.json 2> $OUT/bench_cfg4.log; python -c "import json;print('cfg4', json.load(open('$OUT/bench_cfg4.json'))['ms_per_step'])"
python bench.py --no-cpu-baseline --steps 10 --warmup 3 --autocast-bf16 > $OUT/bench_cfg2_bf16.json 2> $OUT/bench_cfg2_bf16.log; python -c "import json;print('cfg2 bf16', json.load(open('$OUT/bench_cfg2_bf16.json'))['ms_per_step'])"
python bench.py --no-cpu-baseline --steps 5 --warmup 3 --workload cfg5 --autocast-bf16 > $OUT/bench_cfg5_bf16.json 2> $OUT/bench_cfg5_bf16.log; python -c "import json;print('cfg5 bf16', json.load(open('$OUT/bench_cfg5_bf16.json'))['ms_per_step'])"
python tools/kernel_bench.py --cfg 2 --iters 10 > $OUT/kernel_bench_cfg2.txt 2>&1
python tools/kernel_bench.py --cfg 5 --iters 5 > $OUT/kernel_bench_cfg5.txt 2>&1
python tools/kernel_bench.py --norm --cl --iters 10 > $OUT/kernel_bench_norm_cl.txt 2>&1
python tools/gnn_stage_profile.py --cl --iters 10 > $OUT/gnn_stage_profile_fused.txt 2>&1; grep -E "^sum" $OUT/gnn_stage_profile_fused.txt
NEXTOU_PW_FUSE=0 python tools/gnn_stage_profile.py --cl --iters 10 > $OUT/gnn_stage_profile_unfused.txt 2>&1; grep -E "^sum" $OUT/gnn_stage_profile_unfused.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_final -o kt -- python $R/bench.py --steps 4 --warmup 3 --no-cpu-baseline --graph off > $OUT/kt_final_bench.log 2>&1
python $R/tools/rocprof_summary.py /tmp/kt_final $OUT/cfg2_step_kernel_trace_final.md "Round 3 final, cfg 2 train step, eager (rocprofv3 --kernel-trace --stats of python bench.py --steps 4 --warmup 3 --no-cpu-baseline --graph off)" --steady "knn_fused_kernel<28" 2
find /tmp/kt_final -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'head -40 {} > '$OUT'/rocprofv3_kernel_stats_head.csv'
cd $R
bash tools/pmc_traffic.sh "s3 Pool" > $OUT/pmc_traffic_s3pool.log 2>&1
python tools/pmc_table.py gpurun_out/pmc_s3_Pool/FETCH_SIZE gpurun_out/pmc_s3_Pool/WRITE_SIZE > $OUT/pmc_traffic_s3pool.md 2>&1; cat $OUT/pmc_traffic_s3pool.md | cut -c1-250
du -sh $R/gpurun_out
