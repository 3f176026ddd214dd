#!/bin/bash
# Round 4, session m: Winograd dgrad with the XCD-aligned two-region split-K.
set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r4m
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_f16x3.py -q -x > $O/train_tests.log 2>&1
echo "train tests rc $?"; tail -2 $O/train_tests.log
timeout 600 python -m pytest tests/test_gpu_edge.py -q -x -k "hidden_size or emb_size" > $O/shapes.log 2>&1
echo "shape tests rc $?"; tail -2 $O/shapes.log
B="python bench.py --workload train --no-sub --no-cpu-baseline --no-fp32-ref"
for v in 0 1; do
  MV_WINO_DGRAD=$v timeout 300 $B > $O/train_wd$v.json 2> $O/train_wd$v.err
done
python - <<PY
import json
for v in (0, 1):
  try:
    d = json.load(open("$O/train_wd%d.json" % v)); r = d["roofline"]
    print("wino dgrad", v, d["value"], d["ms_per_step"], r.get("per_kernel_ms"), r.get("other_kernels_ms_total"), r["other_kernels_ms"].get("dgrad_slice_sum"))
  except Exception as ex:
    print(v, "failed", ex)
PY
