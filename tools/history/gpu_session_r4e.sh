#!/bin/bash
# Round 4, session e: the Winograd kernel's epilogue through LDS tiles (coalesced state I/O,
# 16-byte plane stores): tests, A/B against the ablation bits, relu / lrelu, full suite.
set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r4e
mkdir -p $O
for w in 8 4; do
  MV_WINO_WAVES=$w timeout 300 python -m pytest tests/test_gpu_wino.py -q -x -s > $O/wino_tests_w$w.log 2>&1
  echo "wino tests waves $w rc $?"; grep -E "winograd  M=2 18x32 Cx=64|passed|failed|Error|assert" $O/wino_tests_w$w.log | tail -8
done
B="python bench.py --no-sub --no-cpu-baseline --no-fp32-ref --steps 40"
for v in "8 0" "4 0" "8 2" "8 12" "8 16" "8 28" "8 32"; do
  set -- $v
  MV_WINO_WAVES=$1 MV_WINO_ABL=$2 timeout 300 $B > $O/greedy_w$1_a$2.json 2> $O/greedy_w$1_a$2.err
done
MV_WINO=0 timeout 300 $B > $O/greedy_direct.json 2> $O/greedy_direct.err
for wl in beam train; do
  timeout 300 $B --workload $wl > $O/${wl}_w8.json 2> $O/${wl}_w8.err
done
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/*.json")):
  try:
    d = json.load(open(f)); r = d["roofline"]
    print(f.split("/")[-1], d["value"], d["ms_per_step"], r.get("avg_launch_ms"), r.get("frac"))
  except Exception as ex:
    print(f, "failed", ex)
PY
timeout 900 python -m pytest tests/test_gpu_edge.py -q -x -s -k "unbounded" > $O/relu.log 2>&1
echo "relu inference rc $?"; grep -E "relu|passed|failed|Error" $O/relu.log | tail -16
timeout 900 python -m pytest tests -m gpu -q -x > $O/gpu_tests.log 2>&1; rc=$?
echo "gpu tests rc $rc"; tail -5 $O/gpu_tests.log
if [ $rc -ne 0 ]; then grep -E "^(FAILED|ERROR)|Error|assert" $O/gpu_tests.log | head -40; fi
