#!/bin/bash
# Round 4: rocm-smi 4x/s under the bf16 greedy forward and the fp32-MFMA greedy forward.
set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r4smi3
mkdir -p $O
B="python bench.py --no-sub --no-cpu-baseline --no-fp32-ref"
sample() { while [ ! -e $O/stop ]; do rocm-smi -c -P 2>/dev/null | grep -E "sclk|Power" >> $1; sleep 0.25; done; }
run() { T=$1; shift; rm -f $O/stop; sample $O/smi_$T.log & SP=$!; timeout 100 $B "$@" > $O/$T.json 2> $O/$T.err; touch $O/stop; wait $SP; }
run bf16 --compute bf16 --scene-conv-kernel 1 --steps 600
run f32 --compute f32 --steps 80
python - <<PY
import json, re
for v in ("bf16", "f32"):
  d = json.load(open("$O/%s.json" % v))
  txt = open("$O/smi_%s.log" % v).read()
  print(v, d["value"], d["ms_per_step"], "sclk", [int(x) for x in re.findall(r"sclk clock level: \d+: \((\d+)Mhz\)", txt)])
  print(v, "power", [float(x) for x in re.findall(r"Power \(W\): ([0-9.]+)", txt)])
PY
