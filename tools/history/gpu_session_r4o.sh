#!/bin/bash
# Round 4, session o: (a) 4- vs 8-wave Winograd workgroups on all three workloads (which default?)
# (b) the stagger experiment by block index.
set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r4o
mkdir -p $O
B="python bench.py --no-sub --no-cpu-baseline --no-fp32-ref"
for w in 8 4; do
  MV_WINO_WAVES=$w timeout 300 $B --steps 60 > $O/greedy_w$w.json 2> $O/greedy_w$w.err
  MV_WINO_WAVES=$w timeout 300 $B --workload beam > $O/beam_w$w.json 2> $O/beam_w$w.err
  MV_WINO_WAVES=$w timeout 300 $B --workload train > $O/train_w$w.json 2> $O/train_w$w.err
done
for s in 25 50 75; do
  MV_WINO_WAVES=4 MV_WINO_STAGGER=$s timeout 300 $B --steps 60 > $O/greedy_w4_st$s.json 2> $O/greedy_w4_st$s.err
done
MV_WINO_WAVES=8 MV_WINO_STAGGER=50 timeout 300 $B --steps 60 > $O/greedy_w8_st50.json 2> $O/greedy_w8_st50.err
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/*.json")):
  try:
    d = json.load(open(f)); r = d["roofline"]
    print(f.split("/")[-1], d["value"], d["ms_per_step"], r.get("avg_launch_ms"), r.get("per_kernel_ms"))
  except Exception as ex:
    print(f, "failed", ex)
PY
