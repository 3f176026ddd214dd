#!/bin/bash
# Round 4, session q: the class decoder's graph attention on a side stream beside the
# hidden2grid / tail launches of the previous step (MV_SIDE_GNN=0: in line): parity tests of the
# greedy forward in stream and graph mode, then the A/B.
# (The side-stream code was removed after this session: zero gain, profiles/r4q_side_stream.txt.)
set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r4q
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_forward.py tests/test_gpu_dropin.py tests/test_gpu_f16x3.py tests/test_gpu_reference_pin.py -q -x -m gpu > $O/tests.log 2>&1
echo "tests rc $?"; tail -3 $O/tests.log
B="python bench.py --no-sub --no-cpu-baseline --no-fp32-ref"
for rep in 1 2; do
for v in 0 1; do
  MV_SIDE_GNN=$v timeout 300 $B --steps 80 > $O/greedy_side${v}_$rep.json 2> $O/greedy_side${v}_$rep.err
  MV_SIDE_GNN=$v timeout 300 $B --steps 80 --graph 1 > $O/greedy_graph_side${v}_$rep.json 2> $O/greedy_graph_side${v}_$rep.err
done
done
for v in 0 1; do
  MV_SIDE_GNN=$v timeout 300 $B --steps 40 --batch 256 > $O/greedy256_side$v.json 2> $O/greedy256_side$v.err
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
