#!/bin/bash
# Round 4, session v (final evidence on the CURRENT kernel sources): the whole GPU suite, the default
# bench line with every sub-workload, rocprofv3 traces + PMC of greedy / beam / train / bf16 train.
set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
T=${1:-r4v}
O=gpurun_out/$T
mkdir -p $O
timeout 1200 python -m pytest tests -q -x -m gpu > $O/gpu_tests.log 2>&1
echo "gpu tests rc $?"; tail -3 $O/gpu_tests.log
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $?"
python - <<PY
import json
d = json.load(open("$O/bench_default.json"))
r = d["roofline"]
print(d["value"], d["ms_per_step"], r["frac"], r.get("avg_launch_ms"), r.get("traffic"), r.get("executed_mfma_frac"))
for k in ("greedy_b256", "greedy_literal_grids", "beam_n128_b20", "train_n32", "bf16", "train_bf16_n64", "fp32_mfma_reference", "cpu_baseline", "host_path"):
  if k in d:
    v = d[k]
    print(k, v.get("value"), v.get("ms_per_step"), (v.get("roofline") or {}).get("frac"))
PY
bash tools/profile_workload.sh ${T}_greedy > $O/prof_greedy.log 2>&1
bash tools/profile_workload.sh ${T}_beam --workload beam > $O/prof_beam.log 2>&1
bash tools/profile_workload.sh ${T}_train --workload train > $O/prof_train.log 2>&1
bash tools/profile_workload.sh ${T}_train_bf16 --workload train --batch 64 --compute bf16 --scene-conv-kernel 1 > $O/prof_train_bf16.log 2>&1
for w in greedy beam train train_bf16; do echo "== $w"; head -8 gpurun_out/prof_${T}_$w/kernel_trace_stats.md; done
python - <<PY
import json, glob
for f in sorted(glob.glob("gpurun_out/prof_${T}_*/pmc_convlstm*.json")):
  d = json.load(open(f))
  print(f.split("/")[-2], d["kernel"], {k: (round(v, 4) if isinstance(v, float) else v) for k, v in d.items() if k not in ("counters", "kernel", "hbm_bytes_per_launch")},
        {k: round(v / 1e6, 1) for k, v in (d.get("hbm_bytes_per_launch") or {}).items()})
PY
