#!/bin/bash
# rocprofv3 evidence for one bench workload (run on the GPU box via gpurun):
#   tools/profile_workload.sh <tag> <bench args...>
# writes gpurun_out/prof_<tag>/{kt,pmc1,pmc2,pmc3}_results.db and the summaries
# gpurun_out/prof_<tag>/*.md|json that get copied into profiles/.
# Counters are collected in their own passes (--pmc with --kernel-trace only).
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
rm -rf $OUT
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python $ROOT/bench.py --no-cpu-baseline --no-sub --no-fp32-ref $*"
rocprofv3 --kernel-trace --stats -d $OUT -o kt -- $B --steps 3 --warmup 1 > $OUT/bench_under_rocprof.json 2> $OUT/kt.err
python $ROOT/tools/rocpd_summary.py $OUT/kt_results.db > $OUT/kernel_trace_stats.md 2>> $OUT/kt.err
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY -d $OUT -o pmc1 -- $B --steps 1 --warmup 1 > /dev/null 2> $OUT/pmc1.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT -o pmc2 -- $B --steps 1 --warmup 1 > /dev/null 2> $OUT/pmc2.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT -o pmc3 -- $B --steps 1 --warmup 1 > /dev/null 2> $OUT/pmc3.err
for k in convlstm_step_kernel convlstm_step_wino3 wino3_transform convlstm_step_wino_kernel convlstm_step_f16x3_lds convlstm_step_bf16 convlstm_dgrad convlstm_wgrad_fast convlstm_wgrad_f16x3 gnn_attend h2g_q decode_tail hidden2grid split_planes lstm_gate_bwd beam_rank beam_select transpose_split wino3_transpose; do
  python $ROOT/tools/pmc_report.py $k $OUT/pmc_$k.json $OUT/pmc1_results.db $OUT/pmc2_results.db $OUT/pmc3_results.db > /dev/null 2>> $OUT/pmc.err
  grep -q '"counters": {}' $OUT/pmc_$k.json && rm -f $OUT/pmc_$k.json
done
rm -f $OUT/*.db
ls $OUT
head -40 $OUT/kernel_trace_stats.md
