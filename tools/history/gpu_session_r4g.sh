#!/bin/bash
# Round 4, session g: wgrad with two stages of load lead: gradient parity, training bench, PMC.
set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r4g
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_f16x3.py -q -x > $O/train_tests.log 2>&1
echo "train tests rc $?"; tail -3 $O/train_tests.log
B="python bench.py --workload train --no-sub --no-cpu-baseline --no-fp32-ref"
timeout 300 $B > $O/train.json 2> $O/train.err
python - <<PY
import json
d = json.load(open("$O/train.json")); r = d["roofline"]
print(d["value"], d["ms_per_step"], r.get("per_kernel_ms"), r.get("other_kernels_ms_total"))
print(r.get("other_kernels_ms"))
PY
bash tools/profile_workload.sh r4g_train --workload train > $O/prof_train.log 2>&1
head -12 gpurun_out/prof_r4g_train/kernel_trace_stats.md
python - <<PY
import json
d = json.load(open("gpurun_out/prof_r4g_train/pmc_convlstm_wgrad_f16x3.json"))
print({x: d[x] for x in d if x != "counters"})
PY
