#!/bin/bash
# Energy attribution of the f16x3 Winograd gate kernel (run on the GPU box through gpurun):
#   tools/energy_attribution.sh <out tag> [steps]
# For every row of the table below: the headline workload (bench.py --no-sub, greedy N = 64,
# random synthetic data) under `rocm-smi -c -P` sampled 4x/s -> ms per grouped launch
# (hipEvents), sclk and package power while running, energy per launch = power x time.
# A row = a library build (build/variants/libmv_r5c<bits>.so = tools/build_variant.sh r5c<bits>
# -DMV_WINO_ABLC=<bits>, main-loop ablations) + a run-time MV_WINO_ABL (epilogue ablations).
# Results of ablated runs are garbage by construction; only time and power are read.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
TAG=${1:-r5a}
STEPS=${2:-500}
O=gpurun_out/$TAG
mkdir -p $O
B="python bench.py --no-sub --no-cpu-baseline --no-fp32-ref --steps $STEPS --warmup 20"
sample() {   # $1 = log; runs until the file $O/stop appears
  while [ ! -e $O/stop ]; do rocm-smi -c -P 2>/dev/null | grep -E "sclk|Power" >> $1; sleep 0.2; done
}
# name : ABLC build bits : run-time MV_WINO_ABL bits
# MV_ENERGY_ROWS overrides the table (same "name:ablc-build:abl" triples; a 4th field "k=v" is
# exported for the run, e.g. f23:0:0:MV_WINO3=0)
ROWS=${MV_ENERGY_ROWS:-"mfma_only:55:2 plus_transform:54:2 plus_dpp:52:2 plus_wreads:48:2 plus_gloads:32:2 plus_dma:0:2 \
plus_epi_math:0:24 plus_state_stores:0:16 shipped:0:0 one_mfma:8:0 no_transform:1:0 no_transc:0:32 idle:-:-"}
for row in $ROWS; do
  IFS=: read -r name c a kv <<< "$row"
  rm -f $O/stop
  sample $O/smi_$name.log &
  SP=$!
  if [ $name = idle ]; then sleep 5
  else
    if [ $c = 0 ]; then unset MV_LIB_PATH; else export MV_LIB_PATH=$ROOT/build/variants/libmv_${MV_ENERGY_LIBPREFIX:-r5c}$c.so; fi
    if [ -n "${kv:-}" ]; then env $kv MV_WINO_ABL=$a timeout 180 $B > $O/$name.json 2> $O/$name.err
    else MV_WINO_ABL=$a timeout 180 $B > $O/$name.json 2> $O/$name.err; fi
  fi
  touch $O/stop; wait $SP
done
unset MV_LIB_PATH
python tools/energy_table.py $O $ROWS | tee $O/table.md
