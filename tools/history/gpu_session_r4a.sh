#!/bin/bash
# Round 4, first GPU session of the Winograd gate kernel: its kernel tests, the new drop-in
# tests, A/B of the two forms on the greedy / beam / train workloads, then the full suite and
# a kernel trace.  Everything lands under gpurun_out/r4a/.
set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r4a
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_wino.py -q -x -s > $O/wino_tests.log 2>&1; rc=$?
echo "wino tests rc $rc"; grep -E "winograd|direct|dynamic|passed|failed" $O/wino_tests.log | tail -30
if [ $rc -ne 0 ]; then tail -60 $O/wino_tests.log; fi
for w in 0 1; do
  MV_WINO=$w timeout 300 python bench.py --no-sub --no-cpu-baseline --no-fp32-ref \
    > $O/greedy_wino$w.json 2> $O/greedy_wino$w.err
  MV_WINO=$w timeout 300 python bench.py --workload beam --no-sub --no-cpu-baseline --no-fp32-ref \
    > $O/beam_wino$w.json 2> $O/beam_wino$w.err
  MV_WINO=$w timeout 300 python bench.py --workload train --no-sub --no-cpu-baseline --no-fp32-ref \
    > $O/train_wino$w.json 2> $O/train_wino$w.err
done
python - <<PY
import json
for wl in ("greedy", "beam", "train"):
  for w in (0, 1):
    try:
      d = json.load(open("$O/%s_wino%d.json" % (wl, w)))
      r = d["roofline"]
      print(wl, "wino", w, d["value"], d["ms_per_step"], r.get("avg_launch_ms"), r.get("frac"),
            r.get("other_kernels_ms_total"))
    except Exception as ex:
      print(wl, w, "failed", ex)
PY
timeout 900 python -m pytest tests -m gpu -q -x > $O/gpu_tests.log 2>&1; rc=$?
echo "gpu tests rc $rc"; tail -5 $O/gpu_tests.log
if [ $rc -ne 0 ]; then grep -E "^(FAILED|ERROR)|Error|assert" $O/gpu_tests.log | head -40; fi
bash tools/profile_workload.sh r4a_greedy > $O/prof_greedy.log 2>&1
head -30 gpurun_out/prof_r4a_greedy/kernel_trace_stats.md
cat gpurun_out/prof_r4a_greedy/pmc_convlstm_step_wino.json 2>/dev/null | head -60
