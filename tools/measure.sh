#!/bin/bash
# One measurement driver for the GPU box (replaces the one-shot tools/r0N_*.sh scripts of rounds 2-4; those are in the git history).
# Run from the repo root on the box:  tools/gpu_retry.sh <timeout> 'bash tools/measure.sh <task> [tag]'
#   tests      full `pytest -m gpu` + smoke, tail of the log
#   margins    the parity tests that print their logit margins (-s), -> <out>/parity_margins.txt
#   bench      headline bench line (cfg 2, with cpu_baseline) + a second line without
#   configs    cfg 4, cfg 2 bf16, cfg 5 bf16 bench lines
#   stages     GNN blocks stand-alone: eager + hipGraph-replay columns, default and NEXTOU_PW_FUSE=0, MIN_POINTS=0
#   kernels    tools/kernel_bench.py at the cfg-2 and cfg-5 shapes, K5 (--bti), K6 (--norm --cl)
#   trace      rocprofv3 --kernel-trace --stats of the eager cfg-2 step -> one steady-state step as a markdown table
#   pmc5       PMC FETCH_SIZE / WRITE_SIZE (separate passes) of the graph kernels at the cfg-5 shapes "s2 Swin" and "s3 Pool"
#   pmcmrg     the same two PMC passes over the K2 + K7 kernel and the three launches it replaces (tools/kernel_bench.py --mrg)
#   pmcjson    regenerate profiles/pmc_traffic.json (-> <out>/pmc_traffic.json) for the labels that launch today
#   convtable  per-layer table of the library convolutions at the model's own routing (tools/conv_layer_table.py)
#   heads      K8 against the library route at the cfg-2 head shapes
#   repro      the guard-page reproducer of MIOpen's backward-data over-read (+ K8 on the same operands)
#   reprobf16  the guard-page reproducer with bf16 operands (tiny and cfg-2 head shapes) + two bf16 bench lines: the reduced-precision fault
#              of round 5's closing tree
#   bf16       round 6: bench.py --autocast-bf16 at cfg 2 / tiny / cfg 5 in find mode, twice each; on a fault a serialized rerun with MIOpen's log
#   bf16ab     round 6: round 5's faulting bf16 configuration (NEXTOU_REDUCED_PRECISION_FILTERS=stored) three times; on a fault once more with synchronous
#              launches and MIOpen's command log (profiles/r06_bf16/README.md)
#   stem       round 6: K9 / DDP tests + the headline line with and without the stem block on one box
#   k1small    round 6: small-graph kNN tests, tools/stem_bench.py, the K1 rows of the kernel bench with and without knn_small_kernel
#   ab6        round 6: same-box A/B of the step changes (K9, small kNN, skip fork) against the round-5 routing (profiles/r06_step_ab.md)
#   ngt1loop   round 6: the default N > 1 mode (two graphs around eager collectives) twenty times on a world-size-1 RCCL group + plain / split / eager lines
#   guard      tests/test_gpu_guard.py + tests/test_gpu_head.py verbose (every own kernel on guard-page operands, outputs and workspaces)
#   glue       the step's small ATen launches by op, shape and enclosing op (tools/aten_glue_profile.py --parents) + bench A/B of the round-5
#              glue changes, hipGraph replay and eager (profiles/r05_aten_glue.md)
#   pwmodes    stage 3-5 blocks with the point-wise convolutions on the library route / on K7 / K7 weight gradient only, replayed graphs
#   stepglue   tests of the step-glue kernels (ClipSGD, narrow_copy_sum) + tools/step_ab.py: the bench step against variants without
#              clip / optimizer and with the own clip + SGD kernels, captured side by side and replayed in alternation
#   optim      tools/optim_bench.py (clip + SGD alone: torch's multi-tensor kernels against ClipSGD) + the RCCL capture / watchdog
#              reproducer (tools/rccl_capture_watchdog_repro.py) + the averaged-step test
#   sgdab      bench.py in separate processes, alternating: default (ClipSGD) against NEXTOU_CLIP_SGD=0 (torch's fused SGD + clip) — the
#              process-level A/B of the optimizer inside the replayed step — and the averaged-step test four times over
#   benchfinal the headline line (default: ClipSGD, one-pass concatenation backward, up-convolutions as GEMMs; with cpu_baseline) and the same
#              step with NEXTOU_CLIP_SGD=0 (torch's clip + fused SGD)
#   averaged   bench.py --force-averager --graph on at cfg 2: the N > 1 step (hooks, buckets, RCCL collectives on a world-size-1 group) at the
#              closing tree, next to the plain step's line
#   cpusurvey  bench.py --cpu-protocol survey (batch 2, 1 + 3 steps, all physical cores; ~10 min of host time)
#   closing    tests margins bench configs stages kernels trace pmc5 pmcmrg in that order
TASK=${1:-closing}
TAG=${2:-r06}
R=$PWD
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_FWD=0 MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_BWD=0 MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_WRW=0
export MIOPEN_LOG_LEVEL=1

field() { python -c "import json,sys;d=json.load(open('$1'));print('$(basename $1)', d['ms_per_step'], 'ms/step', round(d['value']/1e6,2), 'Mvox/s graph', d['config']['step_replayed_as_hipgraph'], d['config']['graph_capture_error'])"; }

t_tests() {
  python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -1 $OUT/smoke.txt
  python -m pytest tests -q -m gpu --durations=10 -rf 2>&1 | grep -v "MIOpen\|GridwiseOp\|amdgpu.ids" > $OUT/pytest_gpu_full_log.txt
  (grep -E "^(FAILED|ERROR)|^E  " $OUT/pytest_gpu_full_log.txt | cut -c1-400 | head -60; tail -26 $OUT/pytest_gpu_full_log.txt) > $OUT/pytest_gpu_full.txt; tail -4 $OUT/pytest_gpu_full.txt
}
t_margins() {
  python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity2.py tests/test_gpu_fused_goldens.py tests/test_gpu_parity_r5.py -q -s -m gpu \
      -k "tiny_models or forward_parity or equal_conv or reference_state_dict or reduced_precision or graph_stack or batch2" 2>&1 | grep -E "max \|dlogit\||mean \|dlogit\||teacher-forced|full-size forward|equal \(fp64\)|passed|failed" > $OUT/parity_margins.txt
  cat $OUT/parity_margins.txt
}
t_bench() {
  python bench.py > $OUT/bench_cfg2_default.json 2> $OUT/bench_cfg2_default.log; field $OUT/bench_cfg2_default.json
  python bench.py --no-cpu-baseline > $OUT/bench_cfg2_again.json 2> $OUT/bench_cfg2_again.log; field $OUT/bench_cfg2_again.json
}
t_configs() {
  python bench.py --no-cpu-baseline --steps 10 --warmup 3 --workload cfg4 > $OUT/bench_cfg4.json 2> $OUT/bench_cfg4.log; field $OUT/bench_cfg4.json
  python bench.py --no-cpu-baseline --steps 10 --warmup 3 --autocast-bf16 > $OUT/bench_cfg2_bf16.json 2> $OUT/bench_cfg2_bf16.log; field $OUT/bench_cfg2_bf16.json
  python bench.py --no-cpu-baseline --steps 5 --warmup 3 --workload cfg5 --autocast-bf16 > $OUT/bench_cfg5_bf16.json 2> $OUT/bench_cfg5_bf16.log; field $OUT/bench_cfg5_bf16.json
}
t_stages() {
  python tools/gnn_stage_profile.py --cl --graph --iters 20 > $OUT/gnn_stage_default.txt 2>&1; grep -E "^sum|^as " $OUT/gnn_stage_default.txt
  NEXTOU_PW_FUSE=0 python tools/gnn_stage_profile.py --cl --graph --iters 20 > $OUT/gnn_stage_unfused.txt 2>&1; grep -E "^sum|^as " $OUT/gnn_stage_unfused.txt
  NEXTOU_PW_FUSE_MIN_POINTS=0 python tools/gnn_stage_profile.py --cl --graph --iters 20 --stages 3,4,5 > $OUT/gnn_stage_min0.txt 2>&1; grep -E "^sum|^as " $OUT/gnn_stage_min0.txt
}
t_kernels() {
  python tools/kernel_bench.py --cfg 2 --iters 10 > $OUT/kernel_bench_cfg2.txt 2>&1
  python tools/kernel_bench.py --cfg 5 --iters 5 > $OUT/kernel_bench_cfg5.txt 2>&1
  python tools/kernel_bench.py --bti --iters 10 > $OUT/kernel_bench_k5.txt 2>&1
  python tools/kernel_bench.py --norm --cl --iters 10 > $OUT/kernel_bench_norm_cl.txt 2>&1
  python tools/kernel_bench.py --mrg --iters 10 > $OUT/kernel_bench_mrg.txt 2>&1
  grep -E "knn_fused|mr_" $OUT/kernel_bench_cfg2.txt | head -24
}
t_stepglue() {
  python -m pytest tests/test_gpu_step_glue.py tests/test_gpu_guard.py tests/test_gpu_ddp.py tests/test_gpu_parity.py tests/test_gpu_parity2.py -q -m gpu -rf \
      -k "step_glue or narrow or clip_sgd or cat_bias or upconv or ddp or tiny_models or train_step or graphed or channels_last_policy or own_bias" 2>&1 \
      | grep -v "MIOpen\|amdgpu.ids" > $OUT/stepglue_pytest_log.txt
  (grep -E "^(FAILED|ERROR)|^E  " $OUT/stepglue_pytest_log.txt | cut -c1-300 | head -40; tail -4 $OUT/stepglue_pytest_log.txt) > $OUT/stepglue_pytest.txt; cat $OUT/stepglue_pytest.txt
  python tools/step_ab.py --rounds 3 --steps 10 --eager > $OUT/step_ab.txt 2> $OUT/step_ab.log; tail -8 $OUT/step_ab.txt; tail -3 $OUT/step_ab.log
}
t_optim() {
  python tools/rccl_capture_watchdog_repro.py 3 > $OUT/rccl_capture_watchdog_repro.txt 2>&1; cat $OUT/rccl_capture_watchdog_repro.txt
  python tools/optim_bench.py 2>&1 | grep -v "amdgpu.ids" > $OUT/optim_bench.txt; cat $OUT/optim_bench.txt
  python -m pytest tests/test_gpu_ddp.py -q -m gpu -rf -k "averaged_step" 2>&1 | grep -v "MIOpen\|amdgpu.ids" | tail -5 > $OUT/ddp_pytest.txt; tail -3 $OUT/ddp_pytest.txt
}
t_sgdab() {
  for i in 1 2; do
    python bench.py --no-cpu-baseline --steps 20 > $OUT/sgdab_own_$i.json 2>/dev/null; field $OUT/sgdab_own_$i.json
    NEXTOU_CLIP_SGD=0 python bench.py --no-cpu-baseline --steps 20 > $OUT/sgdab_torch_$i.json 2>/dev/null; field $OUT/sgdab_torch_$i.json
  done
  for i in 1 2 3 4; do
    python -m pytest tests/test_gpu_ddp.py -q -m gpu -k "averaged_step_over_rccl" 2>&1 | tail -1
  done > $OUT/ddp_repeat.txt; cat $OUT/ddp_repeat.txt
}
t_benchfinal() {
  python bench.py > $OUT/bench_cfg2_default.json 2> $OUT/bench_cfg2_default.log; field $OUT/bench_cfg2_default.json
  python -c "import json;d=json.load(open('$OUT/bench_cfg2_default.json'));print(d['config']['optimizer']);print({k:(v['avg_us'],v['frac']) for k,v in d['roofline_graph'].items() if v and k.startswith('glue')})"
  NEXTOU_CLIP_SGD=0 python bench.py --no-cpu-baseline > $OUT/bench_cfg2_torch_sgd.json 2>/dev/null; field $OUT/bench_cfg2_torch_sgd.json
}
t_averaged() {
  python bench.py --no-cpu-baseline --steps 10 --warmup 3 --force-averager --graph on > $OUT/bench_cfg2_averaged.json 2> $OUT/bench_cfg2_averaged.log; field $OUT/bench_cfg2_averaged.json
  python -c "import json;d=json.load(open('$OUT/bench_cfg2_averaged.json'));print(d['config']['optimizer'], d['dist']['backend'], d['config']['gradient_averager'])"
}
t_trace() {
  cd /tmp && export TMPDIR=/tmp
  rm -rf /tmp/kt_$TAG
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$TAG -o kt -- python $R/bench.py --steps 4 --warmup 3 --no-cpu-baseline --graph off > $OUT/kt_bench.log 2>&1
  python $R/tools/rocprof_summary.py /tmp/kt_$TAG $OUT/cfg2_step_kernel_trace.md "cfg 2 train step, eager (rocprofv3 --kernel-trace --stats of python bench.py --steps 4 --warmup 3 --no-cpu-baseline --graph off)" --steady "knn_fused_kernel<28" 2
  find /tmp/kt_$TAG -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'head -40 {} > '$OUT'/rocprofv3_kernel_stats_head.csv'
  cd $R; head -8 $OUT/cfg2_step_kernel_trace.md | cut -c1-200
}
t_pmc5() {
  cd /tmp && export TMPDIR=/tmp
  for only in "s2 Swin" "s3 Pool"; do
    tag=$(echo "$only" | tr " " "_")
    for c in FETCH_SIZE WRITE_SIZE; do
      rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc5_$tag/$c -o pmc -- python $R/tools/kernel_bench.py --cfg 5 --iters 3 --only "$only" > $OUT/pmc5_${tag}_$c.log 2>&1 || tail -3 $OUT/pmc5_${tag}_$c.log
    done
    python $R/tools/pmc_table.py $OUT/pmc5_$tag/FETCH_SIZE $OUT/pmc5_$tag/WRITE_SIZE > $OUT/pmc5_$tag.md 2>&1
    cat $OUT/pmc5_$tag.md | cut -c1-260
    find $OUT/pmc5_$tag -name "*.csv" -size +2M -delete
  done
  cd $R
}
t_pmcmrg() {
  cd /tmp && export TMPDIR=/tmp
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmcmrg/$c -o pmc -- python $R/tools/kernel_bench.py --mrg --iters 3 > $OUT/pmcmrg_$c.log 2>&1 || tail -3 $OUT/pmcmrg_$c.log
  done
  python $R/tools/pmc_table.py $OUT/pmcmrg/FETCH_SIZE $OUT/pmcmrg/WRITE_SIZE > $OUT/pmcmrg.md 2>&1
  cat $OUT/pmcmrg.md | cut -c1-400
  find $OUT/pmcmrg -name "*.csv" -size +2M -delete
  cd $R
}
t_pmcjson() {
  # profiles/pmc_traffic.json for the launch labels of today's kernels: two counter passes (FETCH_SIZE, WRITE_SIZE: separate, counters only
  # with --kernel-trace) + one plain run that writes the labels, per probe; tools/pmc_traffic_json.py pairs label and kernel by name
  cd /tmp && export TMPDIR=/tmp
  cp $R/profiles/pmc_traffic.json $OUT/pmc_traffic.json
  probe() {   # <tag> <command ...>
    local tag=$1; shift
    "$@" --json $OUT/pmcjson_$tag.json > $OUT/pmcjson_$tag.log 2>&1 || tail -3 $OUT/pmcjson_$tag.log
    for c in FETCH_SIZE WRITE_SIZE; do
      rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmcjson_$tag/$c -o pmc -- "$@" > $OUT/pmcjson_${tag}_$c.log 2>&1 || tail -3 $OUT/pmcjson_${tag}_$c.log
    done
    python $R/tools/pmc_traffic_json.py --labels $OUT/pmcjson_$tag.json --fetch $OUT/pmcjson_$tag/FETCH_SIZE --write $OUT/pmcjson_$tag/WRITE_SIZE \
        --merge $OUT/pmc_traffic.json --drop "knn_fused_kernel<28,2>" "knn_fused_kernel<7,6>" | cut -c1-200
    find $OUT/pmcjson_$tag -name "*.csv" -size +2M -delete
  }
  probe s3pool python $R/tools/kernel_bench.py --cfg 2 --iters 3 --only "s3 Pool"
  probe s2swin python $R/tools/kernel_bench.py --cfg 2 --iters 3 --only "s2 Swin"
  probe mrgswin python $R/tools/kernel_bench.py --mrg --iters 3 --only swin
  probe mrgpool2 python $R/tools/kernel_bench.py --mrg --iters 3 --only "pool s2"
  probe head python $R/tools/head_bench.py --iters 3 --own-only --only "full res"
  probe stem python $R/tools/stem_bench.py --iters 3 --two          # round 6: K9 and K6's two-gradient backward at the stage-0 tensor
  probe s5pool python $R/tools/kernel_bench.py --cfg 2 --iters 3 --only "s5 Pool"      # knn_small_kernel
  probe s5swin python $R/tools/kernel_bench.py --cfg 2 --iters 3 --only "s5 Swin"      # knn_small_kernel at K = 28: the bench line's K1 worst shape
  cd $R
}
t_convtable() {
  python tools/conv_layer_table.py --iters 5 --md $OUT/conv_layer_table.md > $OUT/conv_layer_table.log 2>&1; tail -3 $OUT/conv_layer_table.md | cut -c1-300
}
t_heads() {
  python tools/head_bench.py 2>&1 | grep -v "GridwiseOp\|amdgpu.ids\|MIOpen" > $OUT/head_bench.txt; tail -1 $OUT/head_bench.txt
}
t_repro() {
  python tools/conv_bwd_fault_repro.py --own --repeat 3 --log-dir $OUT/repro_logs 2>&1 | cut -c1-420 > $OUT/conv_bwd_fault_repro.txt; tail -20 $OUT/conv_bwd_fault_repro.txt | cut -c1-160
}
t_reprobf16() {      # next round's first call: the bf16 head convolution on guard pages + the reduced-precision bench lines (known issue)
  python tools/conv_bwd_fault_repro.py --dtype bf16 --repeat 3 --log-dir $OUT/repro_bf16_logs 2>&1 | cut -c1-420 > $OUT/conv_bwd_fault_repro_bf16.txt; tail -20 $OUT/conv_bwd_fault_repro_bf16.txt | cut -c1-160
  python tools/conv_bwd_fault_repro.py --dtype bf16 --size cfg2 --repeat 2 --log-dir $OUT/repro_bf16_logs 2>&1 | cut -c1-420 > $OUT/conv_bwd_fault_repro_bf16_cfg2.txt; tail -20 $OUT/conv_bwd_fault_repro_bf16_cfg2.txt | cut -c1-160
  for i in 1 2; do python bench.py --no-cpu-baseline --steps 10 --warmup 3 --autocast-bf16 > $OUT/bench_cfg2_bf16_$i.json 2> $OUT/bench_cfg2_bf16_$i.log; field $OUT/bench_cfg2_bf16_$i.json || tail -2 $OUT/bench_cfg2_bf16_$i.log; done
}
t_bf16() {           # round 6: the reduced-precision bench path at HEAD — does it complete in find mode?  On a fault: the same run once more with
  # synchronous launches and MIOpen's command log, whose tail names the solver that was running (profiles/r06_bf16_*)
  for w in "cfg2 10" "tiny 5" "cfg5 5"; do set -- $w
    for i in 1 2; do
      timeout 1200 python bench.py --no-cpu-baseline --steps $2 --warmup 3 --workload $1 --autocast-bf16 > $OUT/bench_$1_bf16_$i.json 2> $OUT/bench_$1_bf16_$i.log
      rc=$?; echo "bench $1 bf16 run $i rc $rc"; grep -h "Memory access fault" $OUT/bench_$1_bf16_$i.log | head -2
      if [ $rc -eq 0 ]; then field $OUT/bench_$1_bf16_$i.json; else
        MIOPEN_LOG_LEVEL=6 MIOPEN_ENABLE_LOGGING_CMD=1 HIP_LAUNCH_BLOCKING=1 AMD_SERIALIZE_KERNEL=3 timeout 1500 python bench.py --no-cpu-baseline --steps 2 --warmup 1 --workload $1 --autocast-bf16 2>&1 >/dev/null | grep -v "GridwiseOp\|amdgpu.ids" | tail -c 300000 > $OUT/bench_$1_bf16_${i}_serialized_tail.log
        echo "serialized rerun rc ${PIPESTATUS[0]}"; grep -E "Memory access fault|MIOpenDriver|Solver|solver_id|SolverName" $OUT/bench_$1_bf16_${i}_serialized_tail.log | tail -8 | cut -c1-300
        break
      fi
    done
  done
}
t_bf16ab() {         # round 6: the configuration of round 5's faulting bf16 run (channels-last stored filters reach the library under autocast), 3 runs,
  # then the same with synchronous launches + MIOpen's log if one of them faults: convicts or clears the filter layout
  for i in 1 2 3; do
    NEXTOU_REDUCED_PRECISION_FILTERS=stored timeout 900 python bench.py --no-cpu-baseline --steps 5 --warmup 3 --autocast-bf16 > $OUT/bench_cfg2_bf16_stored_$i.json 2> $OUT/bench_cfg2_bf16_stored_$i.log
    rc=$?; echo "bf16 stored-filter run $i rc $rc"; grep -h "Memory access fault" $OUT/bench_cfg2_bf16_stored_$i.log | head -2
    if [ $rc -ne 0 ]; then
      NEXTOU_REDUCED_PRECISION_FILTERS=stored MIOPEN_LOG_LEVEL=6 MIOPEN_ENABLE_LOGGING_CMD=1 HIP_LAUNCH_BLOCKING=1 AMD_SERIALIZE_KERNEL=3 timeout 1500 python bench.py --no-cpu-baseline --steps 2 --warmup 1 --autocast-bf16 2>&1 >/dev/null | grep -v "GridwiseOp\|amdgpu.ids" | tail -c 300000 > $OUT/bench_cfg2_bf16_stored_serialized_tail.log
      echo "serialized rerun rc ${PIPESTATUS[0]}"; grep -E "Memory access fault|MIOpenDriver|Solver|solver_id|SolverName" $OUT/bench_cfg2_bf16_stored_serialized_tail.log | tail -8 | cut -c1-300
      break
    else field $OUT/bench_cfg2_bf16_stored_$i.json; fi
  done
}
t_stem() {           # round 6: K9 tests + the headline line with and without the stem block on the same box
  python -m pytest tests/test_gpu_stem.py tests/test_gpu_ddp.py -q -m gpu -rf -x 2>&1 | grep -v "MIOpen\|GridwiseOp\|amdgpu.ids" | tail -30 > $OUT/stem_pytest.txt; tail -12 $OUT/stem_pytest.txt
  for i in 1 2; do
    python bench.py --no-cpu-baseline --steps 20 > $OUT/bench_stem_on_$i.json 2> $OUT/bench_stem_on_$i.log; field $OUT/bench_stem_on_$i.json || tail -5 $OUT/bench_stem_on_$i.log
    NEXTOU_STEM_BLOCK=0 python bench.py --no-cpu-baseline --steps 20 > $OUT/bench_stem_off_$i.json 2> $OUT/bench_stem_off_$i.log; field $OUT/bench_stem_off_$i.json
  done
  python -c "
import json
d=json.load(open('$OUT/bench_stem_on_1.json'))
import sys
sys.path.insert(0,'.')
" ; python - <<PYEOF
import json
d=json.load(open('$OUT/bench_stem_on_1.json'))
print('own ms', d['roofline']['own_kernels_ms_per_step'])
PYEOF
}
t_k1small() {        # round 6: the one-launch kernel for a handful of small self graphs — tests, then the K1 rows of the kernel bench with and without it
  python -m pytest tests/test_gpu_knn_small.py tests/test_gpu_stem.py tests/test_gpu_parity.py -q -m gpu -rf -k "knn or stem or small" 2>&1 | grep -v "MIOpen\|GridwiseOp\|amdgpu.ids" | tail -30 > $OUT/k1small_pytest.txt; tail -8 $OUT/k1small_pytest.txt
  python tools/stem_bench.py > $OUT/stem_bench.txt 2>&1; cat $OUT/stem_bench.txt
  python tools/kernel_bench.py --cfg 2 --iters 10 2>&1 | grep -E "knn_" > $OUT/kernel_bench_k1_small.txt; grep -E "^s[45]" $OUT/kernel_bench_k1_small.txt
  NEXTOU_KNN_SMALL=0 python tools/kernel_bench.py --cfg 2 --iters 10 2>&1 | grep -E "knn_" > $OUT/kernel_bench_k1_nosmall.txt; grep -E "^s[45]" $OUT/kernel_bench_k1_nosmall.txt
}
t_ab6() {            # round 6: same-box A/B of the round's step changes — default against K9 / small kNN / skip fork switched off, alternating
  OLD="NEXTOU_STEM_BLOCK=0 NEXTOU_KNN_SMALL=0 NEXTOU_SKIP_FORK=0"
  for i in 1 2; do
    python bench.py --no-cpu-baseline --steps 20 > $OUT/ab6_new_$i.json 2>/dev/null; field $OUT/ab6_new_$i.json
    env $OLD python bench.py --no-cpu-baseline --steps 20 > $OUT/ab6_old_$i.json 2>/dev/null; field $OUT/ab6_old_$i.json
  done
  NEXTOU_SKIP_FORK=0 python bench.py --no-cpu-baseline --steps 20 > $OUT/ab6_nofork.json 2>/dev/null; field $OUT/ab6_nofork.json
  NEXTOU_STEM_BLOCK=0 python bench.py --no-cpu-baseline --steps 20 > $OUT/ab6_nostem.json 2>/dev/null; field $OUT/ab6_nostem.json
}
t_ngt1loop() {       # round 6: the DEFAULT mode of every N > 1 run — two hipGraphs around eager collectives — twenty times over on a world-size-1 RCCL group
  ok=0; for i in $(seq 1 20); do
    timeout 600 python bench.py --steps 3 --warmup 2 --workload tiny --force-averager --no-miopen-find --no-cpu-baseline > $OUT/ngt1_$i.json 2> $OUT/ngt1_$i.log
    rc=$?; if [ $rc -eq 0 ] && python -c "import json;d=json.load(open('$OUT/ngt1_$i.json'));assert d['config']['gradient_averager'] and d['config']['step_replayed_as_hipgraph'] and 'two graphs' in d['config']['graph_mode'] and d['config']['graph_capture_error'] is None and d['dist']['backend']=='nccl'"; then ok=$((ok+1)); rm -f $OUT/ngt1_$i.log; else echo "run $i rc $rc"; tail -3 $OUT/ngt1_$i.log; fi
  done
  echo "default N > 1 mode (two graphs around eager collectives, RCCL world size 1): $ok / 20 clean" | tee $OUT/ngt1_loop.txt
  python bench.py --no-cpu-baseline --force-averager --steps 20 > $OUT/bench_cfg2_averaged_split.json 2> $OUT/bench_cfg2_averaged_split.log; field $OUT/bench_cfg2_averaged_split.json
  python bench.py --no-cpu-baseline --force-averager --graph off --steps 20 > $OUT/bench_cfg2_averaged_eager.json 2> $OUT/bench_cfg2_averaged_eager.log; field $OUT/bench_cfg2_averaged_eager.json
  python bench.py --no-cpu-baseline --steps 20 > $OUT/bench_cfg2_plain_samebox.json 2> /dev/null; field $OUT/bench_cfg2_plain_samebox.json
}
t_guard() {
  python -m pytest tests/test_gpu_guard.py tests/test_gpu_head.py -q -m gpu -p no:cacheprovider 2>&1 | grep -v "MIOpen\|GridwiseOp\|amdgpu.ids" > $OUT/guard_pages_pytest.txt; tail -3 $OUT/guard_pages_pytest.txt
}
t_glue() {
  python tools/aten_glue_profile.py --parents --ops copy_,fill_,add_,flip --top 150 2>&1 | grep -v "amdgpu.ids\|Warning\|warn_once" > $OUT/aten_glue_parents.txt; tail -1 $OUT/aten_glue_parents.txt
  OLD="NEXTOU_STEP_GLUE=0 NEXTOU_CHANNELS_LAST_FILTERS=0 NEXTOU_FILTER_FLIP=0"
  for i in 1 2; do
    python bench.py --no-cpu-baseline --steps 40 > $OUT/glue_graph_new_$i.json 2>/dev/null; field $OUT/glue_graph_new_$i.json
    env $OLD python bench.py --no-cpu-baseline --steps 40 > $OUT/glue_graph_old_$i.json 2>/dev/null; field $OUT/glue_graph_old_$i.json
    python bench.py --no-cpu-baseline --steps 20 --graph off > $OUT/glue_eager_new_$i.json 2>/dev/null; field $OUT/glue_eager_new_$i.json
    env $OLD python bench.py --no-cpu-baseline --steps 20 --graph off > $OUT/glue_eager_old_$i.json 2>/dev/null; field $OUT/glue_eager_old_$i.json
  done
}
t_pwmodes() {
  rm -f $OUT/pw_modes_stages345.txt
  for m in 0 1 wgrad; do
    NEXTOU_PW_GEMM=$m python tools/gnn_stage_profile.py --cl --graph --iters 20 --stages 3,4,5 2>/dev/null | sed "s/^/pw=$m /" >> $OUT/pw_modes_stages345.txt
  done
  grep "as replayed" $OUT/pw_modes_stages345.txt
}
t_cpusurvey() {
  python bench.py --steps 5 --warmup 3 --cpu-protocol survey > $OUT/bench_cfg2_cpu_survey.json 2> $OUT/bench_cfg2_cpu_survey.log
  python -c "import json;d=json.load(open('$OUT/bench_cfg2_cpu_survey.json'));print(d['cpu_baseline'])"
}
case $TASK in
  closing) for t in tests margins bench configs stages kernels heads trace convtable pmcjson repro; do echo "== $t"; t_$t; done ;;
  *) for t in ${TASK//,/ }; do echo "== $t"; t_$t; done ;;
esac
du -sh $OUT | tail -1
