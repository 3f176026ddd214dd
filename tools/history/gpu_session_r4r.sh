#!/bin/bash
# Round 4, session r: the graph attention of step t+1 and hidden2grid of step t as ONE launch
# (post_gate_kernel; MV_POST_GATE=0: separate launches): parity tests, then the A/B.
# (post_gate_kernel was removed after this session: zero gain, profiles/r4r_post_gate_fusion.txt.)
set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r4r
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_forward.py tests/test_gpu_dropin.py tests/test_gpu_f16x3.py tests/test_gpu_reference_pin.py -q -x -m gpu > $O/tests.log 2>&1
echo "tests rc $?"; tail -3 $O/tests.log
B="python bench.py --no-sub --no-cpu-baseline --no-fp32-ref"
for rep in 1 2; do
for v in 0 1; do
  MV_POST_GATE=$v timeout 300 $B --steps 80 > $O/greedy_pg${v}_$rep.json 2> $O/greedy_pg${v}_$rep.err
done
done
for v in 0 1; do
  MV_POST_GATE=$v timeout 300 $B --steps 80 --graph 1 > $O/greedy_graph_pg$v.json 2> $O/greedy_graph_pg$v.err
  MV_POST_GATE=$v timeout 300 $B --steps 40 --batch 256 > $O/greedy256_pg$v.json 2> $O/greedy256_pg$v.err
done
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/*.json")):
  try:
    d = json.load(open(f)); r = d["roofline"]
    print(f.split("/")[-1], d["value"], d["ms_per_step"], r.get("avg_launch_ms"), r.get("other_kernels_ms_total"))
  except Exception as ex:
    print(f, "failed", ex)
PY
