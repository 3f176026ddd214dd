#!/bin/bash
# Training step (configs[2], bench.py --workload train) under a list of env settings, same box:
#   tools/train_ab.sh <tag> "<ENV=V ...>" "<ENV=V ...>" ...      ("-" = the default)
# one line per setting: traj/s, ms per step, ms per gate kernel.
set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
T=$1; shift
O=gpurun_out/$T
mkdir -p $O
i=0
for kv in "$@"; do
  i=$((i + 1))
  f=$O/train_$i.json
  if [ "$kv" = "-" ]; then timeout 300 python bench.py --no-sub --no-cpu-baseline --no-fp32-ref --workload train > $f 2> $O/train_$i.err
  else env $kv timeout 300 python bench.py --no-sub --no-cpu-baseline --no-fp32-ref --workload train > $f 2> $O/train_$i.err; fi
  python - $f "$kv" <<'PY'
import json, sys
try:
  d = json.loads(open(sys.argv[1]).read().strip().split("\n")[-1]); r = d["roofline"]
  print(sys.argv[2], "|", d["value"], "traj/s", d["ms_per_step"], "ms/step", r.get("per_kernel_ms"), "other", r.get("other_kernels_ms_total"))
except Exception as e:
  print(sys.argv[2], "| unreadable:", e)
PY
done
