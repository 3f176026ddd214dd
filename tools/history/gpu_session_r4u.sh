#!/bin/bash
# Round 4, session u: bf16 training after the one-plane transposes; split-K slices of the (now
# three times shorter) bf16 dgrad.
set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r4u
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_bf16.py -q -x -m gpu -s -k training > $O/tests_bf16.log 2>&1
echo "bf16 training test rc $?"; grep -E "passed|failed|error|cosine|bf16 train" $O/tests_bf16.log | tail -4
B="python bench.py --no-sub --no-cpu-baseline --no-fp32-ref --workload train --batch 64 --compute bf16 --scene-conv-kernel 1"
for k in 4 2 1; do
  MV_DGRAD_KSLICES=$k timeout 300 $B > $O/train_bf16_ks$k.json 2> $O/train_bf16_ks$k.err
done
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/*.json")):
  try:
    d = json.load(open(f)); r = d["roofline"]
    o = r.get("other_kernels_ms", {})
    print(f.split("/")[-1], d["value"], d["ms_per_step"], r.get("per_kernel_ms"), r.get("other_kernels_ms_total"),
          "slice_sum", o.get("dgrad_slice_sum"), "transpose", o.get("wgrad_transpose"))
  except Exception as ex:
    print(f, "failed", ex)
PY
