#!/bin/bash
# Round-2 measurement pass, run ON the GPU box from the repo root (gpurun -- 'bash tools/r02_measure.sh').
# Raw rocprofv3 output stays under /tmp; only condensed tables go to gpurun_out/r02/ (64 MiB cap on the way back).
R=$PWD
OUT=$R/gpurun_out/r02
mkdir -p $OUT
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
python -m pytest tests -x -q -m gpu 2>&1 | grep -v "MIOpen\|GridwiseOp" | tail -12 > $OUT/pytest_gpu_full_final.log; tail -3 $OUT/pytest_gpu_full_final.log
python bench.py > $OUT/bench_final.log 2>&1; grep "^{" $OUT/bench_final.log > $OUT/bench_final.json; cut -c1-260 $OUT/bench_final.json
python tools/kernel_bench.py --cfg 2 --iters 10 > $OUT/kernel_bench_cfg2_final.txt 2>&1
NEXTOU_MR_THREADS=256 python tools/kernel_bench.py --cfg 2 --iters 10 --only Pool > $OUT/kernel_bench_cfg2_pool_256thr.txt 2>&1
python tools/kernel_bench.py --cfg 5 --iters 10 > $OUT/kernel_bench_cfg5_final.txt 2>&1
python tools/kernel_bench.py --norm --cl --iters 10 > $OUT/kernel_bench_norm_cl_final.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_final -o kt -- python $R/bench.py --steps 4 --warmup 3 --no-cpu-baseline > $OUT/kt_final_bench.log 2>&1
python $R/tools/rocprof_summary.py /tmp/kt_final $OUT/cfg2_step_kernel_trace_final.md "Round 2 final, cfg 2 train step (rocprofv3 --kernel-trace --stats)" --steady "knn_fused_kernel<28" 2
find /tmp/kt_final -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'head -40 {} > '$OUT'/rocprofv3_kernel_stats_head.csv'
# HBM traffic of the dominant own kernel (K6 on the padded stage-0 tensor): FETCH_SIZE and WRITE_SIZE in separate passes
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_k6_$c -o pmc -- python $R/tools/kernel_bench.py --norm --cl --iters 3 --only "s0 padded" > $OUT/pmc_k6_$c.log 2>&1
done
python $R/tools/pmc_table.py /tmp/pmc_k6_FETCH_SIZE /tmp/pmc_k6_WRITE_SIZE --match bn_cl > $OUT/pmc_traffic_k6_padded.md 2>&1
# SQ counters of the K2 kernels at the Swin s2 shape
for grp in "SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/sq_k2_$i -o pmc -- python $R/tools/kernel_bench.py --cfg 2 --iters 3 --only "s2 Swin" > $OUT/sq_k2_$i.log 2>&1
done
python $R/tools/pmc_table.py /tmp/sq_k2_1 /tmp/sq_k2_2 --match "nextou::" > $OUT/sq_counters_k2_swin_s2.md 2>&1
# bf16: what MIOpen picks in NDHWC vs NCDHW
NEXTOU_REDUCED_PRECISION_LAYOUT=ncdhw rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_bf16_ncdhw -o kt -- python $R/bench.py --steps 3 --warmup 3 --no-cpu-baseline --autocast-bf16 > $OUT/kt_bf16_ncdhw.log 2>&1
python $R/tools/rocprof_summary.py /tmp/kt_bf16_ncdhw $OUT/bf16_trace_ncdhw.md "cfg 2 under bf16 autocast, round-1 policy (NEXTOU_REDUCED_PRECISION_LAYOUT=ncdhw: NCDHW, no padding)" --steady "knn_fused_kernel<28" 2
rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_bf16_ndhwc -o kt -- python $R/bench.py --steps 3 --warmup 3 --no-cpu-baseline --autocast-bf16 > $OUT/kt_bf16_ndhwc.log 2>&1
python $R/tools/rocprof_summary.py /tmp/kt_bf16_ndhwc $OUT/bf16_trace_ndhwc.md "cfg 2 under bf16 autocast, default policy (every stage NDHWC + channel padding)" --steady "knn_fused_kernel<28" 2
grep "^{" $OUT/kt_bf16_ncdhw.log | cut -c1-200; grep "^{" $OUT/kt_bf16_ndhwc.log | cut -c1-200
du -sh $R/gpurun_out
