#!/bin/bash
# Round 4, session k: hand-scheduled main loop of the Winograd kernel (MV_WINO_SCHED=1): tests + A/B.
set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r4k
mkdir -p $O
for w in 8 4; do
  MV_WINO_SCHED=1 MV_WINO_WAVES=$w timeout 300 python -m pytest tests/test_gpu_wino.py -q -x > $O/wino_tests_sched_w$w.log 2>&1
  echo "wino tests sched waves $w rc $?"; tail -2 $O/wino_tests_sched_w$w.log
done
B="python bench.py --no-sub --no-cpu-baseline --no-fp32-ref --steps 40"
for v in "8 0" "8 1" "4 0" "4 1"; do
  set -- $v
  MV_WINO_WAVES=$1 MV_WINO_SCHED=$2 timeout 300 $B > $O/greedy_w$1_s$2.json 2> $O/greedy_w$1_s$2.err
done
MV_WINO_SCHED=1 MV_WINO_ABL=2 timeout 300 $B > $O/greedy_w8_s1_a2.json 2> $O/greedy_w8_s1_a2.err
MV_WINO_SCHED=0 MV_WINO_ABL=2 timeout 300 $B > $O/greedy_w8_s0_a2.json 2> $O/greedy_w8_s0_a2.err
MV_WINO_SCHED=1 timeout 300 $B --workload beam > $O/beam_w8_s1.json 2> $O/beam_w8_s1.err
MV_WINO_SCHED=0 timeout 300 $B --workload beam > $O/beam_w8_s0.json 2> $O/beam_w8_s0.err
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/*.json")):
  try:
    d = json.load(open(f)); r = d["roofline"]
    print(f.split("/")[-1], d["value"], d["ms_per_step"], r.get("avg_launch_ms"), r.get("frac"))
  except Exception as ex:
    print(f, "failed", ex)
PY
