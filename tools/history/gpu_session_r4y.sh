#!/bin/bash
# Round 4, session y: the tree as committed -- smoke(), the kernel tests, and the default bench line,
# which now quotes roofline.traffic from profiles/r4v_* (same kernel-source hash).
set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r4y
mkdir -p $O
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc $?"; tail -2 $O/smoke.log
timeout 600 python -m pytest tests/test_gpu_wino.py tests/test_gpu_kernels.py tests/test_gpu_forward.py -q -x -m gpu > $O/tests.log 2>&1
echo "tests rc $?"; tail -2 $O/tests.log
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $?"
python - <<PY
import json
d = json.load(open("$O/bench_default.json"))
r = d["roofline"]
print(d["value"], d["ms_per_step"], r["frac"], r.get("avg_launch_ms"), r.get("traffic"))
for k in ("greedy_b256", "greedy_literal_grids", "beam_n128_b20", "train_n32", "bf16", "train_bf16_n64"):
  v = d[k]; print(k, v.get("value"), v.get("ms_per_step"), (v.get("roofline") or {}).get("frac"), (v.get("roofline") or {}).get("traffic"))
PY
