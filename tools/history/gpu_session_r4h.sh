#!/bin/bash
# Round 4, session h: cell-state tile prefetched by LDS-DMA: kernel tests, greedy / beam bench,
# graph + beam + forward tests.
set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r4h
mkdir -p $O
for w in 8 4; do
  MV_WINO_WAVES=$w timeout 300 python -m pytest tests/test_gpu_wino.py -q -x > $O/wino_tests_w$w.log 2>&1
  echo "wino tests waves $w rc $?"; tail -2 $O/wino_tests_w$w.log
done
B="python bench.py --no-sub --no-cpu-baseline --no-fp32-ref --steps 40"
for v in "8 0" "4 0" "8 4" "8 2"; do
  set -- $v
  MV_WINO_WAVES=$1 MV_WINO_ABL=$2 timeout 300 $B > $O/greedy_w$1_a$2.json 2> $O/greedy_w$1_a$2.err
done
MV_WINO=0 timeout 300 $B > $O/greedy_direct.json 2> $O/greedy_direct.err
timeout 300 $B --workload beam > $O/beam_w8.json 2> $O/beam_w8.err
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/*.json")):
  try:
    d = json.load(open(f)); r = d["roofline"]
    print(f.split("/")[-1], d["value"], d["ms_per_step"], r.get("avg_launch_ms"), r.get("frac"))
  except Exception as ex:
    print(f, "failed", ex)
PY
timeout 900 python -m pytest tests/test_gpu_forward.py tests/test_gpu_edge.py tests/test_gpu_dropin.py -q -x > $O/fwd_tests.log 2>&1
echo "forward / edge / dropin rc $?"; tail -3 $O/fwd_tests.log
