#!/bin/bash
# Round 4, session l: dgrad in Winograd form: gradient parity, training A/B.
set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r4l
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_f16x3.py -q -x -s > $O/train_tests.log 2>&1
echo "train tests rc $?"; grep -E "passed|failed|Error|assert" $O/train_tests.log | tail -6
B="python bench.py --workload train --no-sub --no-cpu-baseline --no-fp32-ref"
for v in 0 1; do
  MV_WINO_DGRAD=$v timeout 300 $B > $O/train_wd$v.json 2> $O/train_wd$v.err
done
python - <<PY
import json
for v in (0, 1):
  try:
    d = json.load(open("$O/train_wd%d.json" % v)); r = d["roofline"]
    print("wino dgrad", v, d["value"], d["ms_per_step"], r.get("per_kernel_ms"), r.get("other_kernels_ms_total"))
  except Exception as ex:
    print(v, "failed", ex)
PY
timeout 900 python -m pytest tests/test_gpu_train_variants.py tests/test_gpu_reference_pin.py -q -x > $O/variants.log 2>&1
echo "variants + reference pins rc $?"; tail -3 $O/variants.log
