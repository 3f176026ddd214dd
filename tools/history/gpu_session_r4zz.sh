#!/bin/bash
# Round 4, last session: the default bench line from the committed tree with the counter traffic of
# the beam / training sub-lines quoted from profiles/r4z_* (bench.py quote_sub_traffic).
set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r4zz
mkdir -p $O
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err; echo "bench rc $?"
python - <<PY
import json
d = json.load(open("$O/bench_driver_cmd.json"))
r = d["roofline"]
print(d["value"], d["ms_per_step"], r["frac"], r.get("avg_launch_ms"), r.get("traffic"), r.get("traffic_source"))
for k in ("greedy_b256", "greedy_literal_grids", "beam_n128_b20", "train_n32", "bf16", "train_bf16_n64"):
  v = d[k]; rr = v.get("roofline") or {}
  print(k, v.get("value"), v.get("ms_per_step"), rr.get("frac"), rr.get("traffic"), rr.get("traffic_note"), sorted((rr.get("traffic_per_kernel") or {}).keys()))
PY
