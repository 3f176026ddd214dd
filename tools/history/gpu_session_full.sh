#!/bin/bash
# One GPU session after a kernel change (run through gpurun from the repo root):
#   gate tests of the changed kernel -> full GPU suite -> A/B of the old and new graph
#   attention on the beam workload -> rocprofv3 evidence for the four bench workloads ->
#   the default bench line.  Everything lands under gpurun_out/<tag>/ and gpurun_out/prof_r3_*.
set -u
TAG=${1:-session}
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/$TAG
mkdir -p $O
rm -rf gpurun_out/prof_r3_*
timeout 400 python -m pytest tests/test_gpu_kernels.py -q -x -k gnn > $O/gnn_tests.log 2>&1; rc=$?
echo "gnn tests rc $rc"; tail -3 $O/gnn_tests.log
if [ $rc -ne 0 ]; then tail -80 $O/gnn_tests.log; exit 1; fi
timeout 900 python -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1; rc=$?
echo "gpu tests rc $rc"; tail -4 $O/gpu_tests.log
if [ $rc -ne 0 ]; then grep -E "^(FAILED|ERROR)" $O/gpu_tests.log; fi     # measurements follow either way
for v in ${AB_GNN:-v2 v3}; do      # AB_GNN=" " skips the A/B
  MV_GNN=$v timeout 200 python bench.py --workload beam --no-sub --no-cpu-baseline --no-fp32-ref \
    > $O/beam_gnn_$v.json 2> $O/beam_gnn_$v.err
  MV_GNN=$v timeout 200 python bench.py --no-sub --no-cpu-baseline --no-fp32-ref \
    > $O/greedy_gnn_$v.json 2> $O/greedy_gnn_$v.err
done
python - <<PY
import json
for w in ("beam", "greedy"):
  for v in "${AB_GNN:-v2 v3}".split():
    try:
      d = json.load(open("$O/%s_gnn_%s.json" % (w, v)))
      r = d["roofline"]
      print(w, v, d["value"], d["ms_per_step"], r.get("avg_launch_ms"), r.get("other_kernels_ms_total"),
            (r.get("hbm_kernels") or {}).get("gnn_attend"))
    except Exception as ex:
      print(w, v, "failed", ex)
PY
bash tools/profile_workload.sh r3_greedy > $O/prof_greedy.log 2>&1
bash tools/profile_workload.sh r3_beam --workload beam > $O/prof_beam.log 2>&1
bash tools/profile_workload.sh r3_train --workload train > $O/prof_train.log 2>&1
bash tools/profile_workload.sh r3_greedy_bf16 --compute bf16 > $O/prof_bf16.log 2>&1
grep -h "gnn_attend" gpurun_out/prof_r3_*/kernel_trace_stats.md
timeout 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $?"
python - <<PY
import json
d = json.load(open("$O/bench_default.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("traffic"))
for k in ("greedy_b256", "beam_n128_b20", "train_n32", "bf16", "train_bf16_n64", "greedy_literal_grids"):
  if k in d: print(k, d[k]["value"], d[k]["ms_per_step"])
PY
