#!/bin/bash
# Round 4, last session: rocm-smi (sclk, package power) sampled 4x/s while the headline workload runs --
# is the Winograd gate kernel at the package power cap as the direct kernel was (profiles/r3_gate_kernel_power_bound.md)?
set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r4smi
mkdir -p $O
B="python bench.py --no-sub --no-cpu-baseline --no-fp32-ref"
sample() {   # $1 = log; runs until the file $O/stop appears
  while [ ! -e $O/stop ]; do rocm-smi -c -P 2>/dev/null | grep -E "sclk|Power" >> $1; sleep 0.25; done
}
for v in wino direct; do
  rm -f $O/stop
  sample $O/smi_$v.log &
  SP=$!
  if [ $v = wino ]; then timeout 120 $B --steps 500 > $O/greedy_$v.json 2> $O/greedy_$v.err
  else MV_WINO=0 timeout 120 $B --steps 500 > $O/greedy_$v.json 2> $O/greedy_$v.err; fi
  touch $O/stop; wait $SP
done
python - <<PY
import json, re
for v in ("wino", "direct"):
  d = json.load(open("$O/greedy_%s.json" % v)); r = d["roofline"]
  txt = open("$O/smi_%s.log" % v).read()
  sclk = [int(x) for x in re.findall(r"sclk clock level: \d+: \((\d+)Mhz\)", txt)]
  pw = [float(x) for x in re.findall(r"Power \(W\): ([0-9.]+)", txt)]
  busy = lambda a: sorted(a)[len(a) // 2:] if a else []
  print(v, d["value"], r["avg_launch_ms"], "samples", len(sclk), len(pw),
        "sclk upper-half range", (min(busy(sclk)), max(busy(sclk))) if sclk else None,
        "power upper-half range", (min(busy(pw)), max(busy(pw))) if pw else None)
PY
tail -4 $O/smi_wino.log
