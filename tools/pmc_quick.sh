#!/bin/bash
# Ad-hoc PMC passes for one kernel of a bench workload (run on the GPU box):
#   tools/pmc_quick.sh <tag> <kernel substring> <bench args ...> -- "<counters of pass 1>" ["<pass 2>" ...]
# Each pass = rocprofv3 --kernel-trace --pmc <counters> (nothing else: gpurun's rule); prints
# per counter: total per launch, and per CU-cycle (GRBM_GUI_ACTIVE of the same pass / 8 XCDs
# x 256 CUs) where that pass has GRBM_GUI_ACTIVE.
set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
TAG=$1; K=$2; shift 2
BA=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do BA+=("$1"); shift; done
shift
OUT=$PWD/gpurun_out/pmcq_$TAG
mkdir -p $OUT
i=0
for ctrs in "$@"; do
  i=$((i + 1))
  rocprofv3 --kernel-trace --pmc $ctrs -d $OUT -o p$i -- python bench.py --no-sub --no-cpu-baseline --no-fp32-ref "${BA[@]}" --steps 1 --warmup 1 > /dev/null 2> $OUT/p$i.err
  python - $OUT/p${i}_results.db "$K" <<'PY'
import sqlite3, sys
try:
  c = sqlite3.connect(sys.argv[1])
  rows = c.execute("select counter_name, sum(value), count(*), sum(duration) from counters_collection "
                   "where kernel_name like ? group by counter_name", ("%" + sys.argv[2] + "%",)).fetchall()
except Exception as e:
  print("pass unreadable:", e); sys.exit(0)
d = {n: (t, k, dur) for n, t, k, dur in rows}
cyc = d["GRBM_GUI_ACTIVE"][0] / 8.0 if "GRBM_GUI_ACTIVE" in d else None
for n, (t, k, dur) in sorted(d.items()):
  s = "%-28s launches %4d  per launch %.4g  avg us %.1f" % (n, k, t / k, dur / k / 1e3)
  if cyc: s += "  per CU-cycle %.4f  per SIMD-cycle %.4f" % (t / (cyc * 256), t / (cyc * 1024))
  print(s)
PY
  tail -2 $OUT/p$i.err | cut -c1-200
done
