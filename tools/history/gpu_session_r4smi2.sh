#!/bin/bash
# Round 4: rocm-smi (sclk, package power) 4x/s under the training step (f16x3, 32 per GPU), the bf16
# training step (64 per GPU) and the beam-20 decode -- which of them sit at the package power cap?
set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r4smi2
mkdir -p $O
B="python bench.py --no-sub --no-cpu-baseline --no-fp32-ref"
sample() { while [ ! -e $O/stop ]; do rocm-smi -c -P 2>/dev/null | grep -E "sclk|Power" >> $1; sleep 0.25; done; }
run() {   # $1 = tag, rest = bench args
  T=$1; shift
  rm -f $O/stop
  sample $O/smi_$T.log &
  SP=$!
  timeout 100 $B "$@" > $O/$T.json 2> $O/$T.err
  touch $O/stop; wait $SP
}
run train --workload train --steps 120
run train_bf16 --workload train --batch 64 --compute bf16 --scene-conv-kernel 1 --steps 100
run beam --workload beam --steps 30
python - <<PY
import json, re
for v in ("train", "train_bf16", "beam"):
  d = json.load(open("$O/%s.json" % v)); r = d["roofline"]
  txt = open("$O/smi_%s.log" % v).read()
  sclk = [int(x) for x in re.findall(r"sclk clock level: \d+: \((\d+)Mhz\)", txt)]
  pw = [float(x) for x in re.findall(r"Power \(W\): ([0-9.]+)", txt)]
  print(v, d["value"], d["ms_per_step"], "sclk", sclk)
  print(v, "power", pw)
PY
