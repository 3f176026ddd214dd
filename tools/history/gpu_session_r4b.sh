#!/bin/bash
# Round 4, session b: variants of the Winograd gate kernel (waves per workgroup, pipelined
# transform), timing ablations, the two-rank in-library all-reduce test.
set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r4b
mkdir -p $O
for v in "4 0" "8 1" "4 1"; do
  set -- $v
  MV_WINO_WAVES=$1 MV_WINO_PIPE=$2 timeout 300 python -m pytest tests/test_gpu_wino.py -q -x > $O/wino_tests_w$1_p$2.log 2>&1
  echo "wino tests waves $1 pipe $2 rc $?"; tail -2 $O/wino_tests_w$1_p$2.log
done
timeout 600 python -m pytest tests/test_gpu_parallel.py -q -x -s -k "two_ranks" > $O/parallel.log 2>&1
echo "parallel rc $?"; grep -E "reduced|parameters after|passed|failed|Error" $O/parallel.log | tail
B="python bench.py --no-sub --no-cpu-baseline --no-fp32-ref"
for v in "8 0 0" "4 0 0" "8 1 0" "4 1 0" "8 0 1" "8 0 2" "4 0 2"; do
  set -- $v
  MV_WINO_WAVES=$1 MV_WINO_PIPE=$2 MV_WINO_ABL=$3 timeout 300 $B > $O/greedy_w$1_p$2_a$3.json 2> $O/greedy_w$1_p$2_a$3.err
done
for v in "8 0 0" "4 0 0" "4 1 0"; do
  set -- $v
  MV_WINO_WAVES=$1 MV_WINO_PIPE=$2 timeout 300 $B --workload beam > $O/beam_w$1_p$2_a$3.json 2> $O/beam_w$1_p$2_a$3.err
done
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/*_w*_p*_a*.json")):
  try:
    d = json.load(open(f)); r = d["roofline"]
    print(f.split("/")[-1], d["value"], d["ms_per_step"], r.get("avg_launch_ms"), r.get("frac"))
  except Exception as ex:
    print(f, "failed", ex)
PY
