#!/bin/bash
# Round 4, session d: where the epilogue of the Winograd gate kernel spends its time (MV_WINO_ABL
# bits, 8-wave and 4-wave forms); emb_size tests.
set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r4d
mkdir -p $O
B="python bench.py --no-sub --no-cpu-baseline --no-fp32-ref --steps 40"
for w in 8 4; do
for a in 0 2 4 8 16 32 12 28 60; do
  MV_WINO_WAVES=$w MV_WINO_ABL=$a timeout 300 $B > $O/greedy_w${w}_a$a.json 2> $O/greedy_w${w}_a$a.err
done
done
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/greedy_w*_a*.json")):
  try:
    d = json.load(open(f)); r = d["roofline"]
    print(f.split("/")[-1], d["value"], d["ms_per_step"], r.get("avg_launch_ms"), r.get("frac"))
  except Exception as ex:
    print(f, "failed", ex)
PY
timeout 900 python -m pytest tests/test_gpu_edge.py -q -x -s -k "emb_size or rejects" > $O/emb.log 2>&1
echo "emb tests rc $?"; grep -E "emb_size|passed|failed|Error" $O/emb.log | tail -20
timeout 900 python -m pytest tests/test_gpu_edge.py -q -x -s -k "unbounded" > $O/relu.log 2>&1
echo "relu inference rc $?"; grep -E "relu|passed|failed|Error" $O/relu.log | tail -12
timeout 900 python -m pytest tests/test_gpu_train_variants.py -q -x -k "relu" > $O/relu_train.log 2>&1
echo "relu train variants rc $?"; tail -3 $O/relu_train.log
