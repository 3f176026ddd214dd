#!/bin/bash
# Round 4, session w: split-K count of the wgrad GEMMs (MV_WGRAD_SPLITS; default 24 at these sizes)
# for the f16x3 training step at 32 per GPU and the bf16 one at 64.
set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r4w
mkdir -p $O
B="python bench.py --no-sub --no-cpu-baseline --no-fp32-ref --workload train"
for k in 24 8 16 32 48; do
  MV_WGRAD_SPLITS=$k timeout 200 $B > $O/train_f16x3_s$k.json 2> $O/train_f16x3_s$k.err
done
for k in 24 16 32 48 64; do
  MV_WGRAD_SPLITS=$k timeout 200 $B --batch 64 --compute bf16 --scene-conv-kernel 1 > $O/train_bf16_s$k.json 2> $O/train_bf16_s$k.err
done
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/*.json")):
  try:
    d = json.load(open(f)); r = d["roofline"]
    o = r.get("other_kernels_ms", {})
    print(f.split("/")[-1], d["value"], d["ms_per_step"], r.get("per_kernel_ms"), r.get("other_kernels_ms_total"), "reduce", o.get("wgrad_reduce"))
  except Exception as ex:
    print(f, "failed", ex)
PY
