#!/bin/bash
# Round 4, session f: DMA-staged f16x3 wgrad kernel: gradient parity tests, A/B on the training
# workload, kernel trace.
set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r4f
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_f16x3.py -q -x -s > $O/train_tests.log 2>&1
echo "train tests rc $?"; grep -E "passed|failed|Error|assert" $O/train_tests.log | tail -8
B="python bench.py --workload train --no-sub --no-cpu-baseline --no-fp32-ref"
for v in v1 v2; do
  MV_WGRAD=$v timeout 300 $B > $O/train_wgrad_$v.json 2> $O/train_wgrad_$v.err
done
python - <<PY
import json
for v in ("v1", "v2"):
  try:
    d = json.load(open("$O/train_wgrad_%s.json" % v)); r = d["roofline"]
    print(v, d["value"], d["ms_per_step"], r.get("per_kernel_ms"), r.get("other_kernels_ms_total"))
  except Exception as ex:
    print(v, "failed", ex)
PY
timeout 600 python -m pytest tests/test_gpu_at_size.py -q -x -k "train" > $O/at_size_train.log 2>&1
echo "at-size train rc $?"; tail -3 $O/at_size_train.log
bash tools/profile_workload.sh r4f_train --workload train > $O/prof_train.log 2>&1
head -24 gpurun_out/prof_r4f_train/kernel_trace_stats.md
python - <<PY
import json
for k in ("convlstm_wgrad_f16x3",):
  try:
    d = json.load(open("gpurun_out/prof_r4f_train/pmc_%s.json" % k))
    print(k, {x: d[x] for x in d if x != "counters"})
  except Exception as ex:
    print(k, ex)
PY
