#!/bin/bash
# Round 4, session c: 4-wave Winograd workgroups with the second workgroup of a CU staggered,
# the XCD-pair column-block map; emb_size tests; parity of the chosen variant.
set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r4c
mkdir -p $O
B="python bench.py --no-sub --no-cpu-baseline --no-fp32-ref"
for v in "8 0 0" "4 0 0" "4 30 0" "4 50 0" "4 70 0" "4 50 1" "4 0 1" "8 0 1"; do
  set -- $v
  MV_WINO_WAVES=$1 MV_WINO_STAGGER=$2 MV_WINO_MAP=$3 timeout 300 $B > $O/greedy_w$1_s$2_m$3.json 2> $O/greedy_w$1_s$2_m$3.err
done
for v in "4 0 0" "4 50 0" "4 50 1"; do
  set -- $v
  MV_WINO_WAVES=$1 MV_WINO_STAGGER=$2 MV_WINO_MAP=$3 timeout 300 $B --workload beam > $O/beam_w$1_s$2_m$3.json 2> $O/beam_w$1_s$2_m$3.err
done
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/*_w*_s*_m*.json")):
  try:
    d = json.load(open(f)); r = d["roofline"]
    print(f.split("/")[-1], d["value"], d["ms_per_step"], r.get("avg_launch_ms"), r.get("frac"))
  except Exception as ex:
    print(f, "failed", ex)
PY
MV_WINO_WAVES=4 MV_WINO_STAGGER=50 MV_WINO_MAP=1 timeout 300 python -m pytest tests/test_gpu_wino.py -q -x > $O/wino_tests.log 2>&1
echo "wino tests (4, 50, 1) rc $?"; tail -2 $O/wino_tests.log
timeout 900 python -m pytest tests/test_gpu_edge.py -q -x -s -k "emb_size or rejects" > $O/emb.log 2>&1
echo "emb tests rc $?"; grep -E "emb_size|passed|failed|Error" $O/emb.log | tail -20
