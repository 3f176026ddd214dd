#!/bin/bash
# A/B builds of the library for tuning sessions: tools/build_variant.sh <name> [-Dflags...]
# -> build/variants/libmv_<name>.so, selected at run time with MV_LIB_PATH (same C ABI).
set -eu
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
mkdir -p $ROOT/build/variants
exec /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off \
  "$@" $ROOT/multiverse_amd/csrc/engine.hip -o $ROOT/build/variants/libmv_$NAME.so
