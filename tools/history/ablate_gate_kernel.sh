#!/bin/bash
# Ablation of the f16x3 gate kernel's main loop on the real launch geometry (run on the GPU
# box):  tools/ablate_gate_kernel.sh <out tag> <variant names...>
# Every variant is a build/variants/libmv_<name>.so (tools/build_variant.sh <name> -DMV_ABL=n)
# run through bench.py's headline workload (N=64 greedy, random synthetic data): one
# un-profiled pass for the hipEvent launch time, one rocprofv3 --pmc pass for MFMA busy,
# effective clock and the wave-cycle split.  Results of MV_ABL != 0 builds are garbage by
# construction; only their timing is read.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  export MV_LIB_PATH=$ROOT/build/variants/libmv_$v.so
  [ -f $MV_LIB_PATH ] || { echo "missing $MV_LIB_PATH"; continue; }
  B="python $ROOT/bench.py --no-sub --no-cpu-baseline --no-fp32-ref"
  timeout 300 $B --steps 30 --warmup 3 > $OUT/$v.bench.json 2> $OUT/$v.bench.err
  rm -rf $OUT/pmc_$v
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY -d $OUT/pmc_$v -o pmc -- $B --steps 2 --warmup 1 > /dev/null 2> $OUT/$v.pmc.err
  python $ROOT/tools/pmc_report.py convlstm_step_f16x3_lds $OUT/$v.pmc.json $OUT/pmc_$v/pmc_results.db > /dev/null 2>> $OUT/$v.pmc.err
  rm -rf $OUT/pmc_$v
done
unset MV_LIB_PATH
python $ROOT/tools/ablate_table.py $OUT "$@" | tee $OUT/table.md
