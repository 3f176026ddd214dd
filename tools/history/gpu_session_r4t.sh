#!/bin/bash
# Round 4, session t: the backward of the bf16 compute mode on one plane per operand (bf16 dgrad,
# wgrad on the leading fp16 planes) against the f16x3 backward it had (MV_BF16_BWD=0).
set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r4t
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_bf16.py -q -x -m gpu -s > $O/tests_bf16.log 2>&1
echo "bf16 tests rc $?"; grep -E "passed|failed|error|cosine|bf16 train" $O/tests_bf16.log | tail -8
MV_BF16_BWD=0 timeout 600 python -m pytest tests/test_gpu_bf16.py -q -x -m gpu -s -k training > $O/tests_bf16_old.log 2>&1
echo "bf16 tests (f16x3 backward) rc $?"; grep -E "passed|failed|error|cosine|bf16 train" $O/tests_bf16_old.log | tail -4
timeout 600 python -m pytest tests/test_gpu_train.py tests/test_gpu_edge.py -q -x -m gpu -k "bf16 or mode" > $O/tests_more.log 2>&1
echo "more tests rc $?"; tail -2 $O/tests_more.log
B="python bench.py --no-sub --no-cpu-baseline --no-fp32-ref --workload train --batch 64 --compute bf16 --scene-conv-kernel 1"
MV_BF16_BWD=0 timeout 300 $B > $O/train_bf16_bwd0.json 2> $O/train_bf16_bwd0.err
timeout 300 $B > $O/train_bf16_bwd1.json 2> $O/train_bf16_bwd1.err
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/*.json")):
  try:
    d = json.load(open(f)); r = d["roofline"]
    print(f.split("/")[-1], d["value"], d["ms_per_step"], r.get("per_kernel_ms"), r.get("other_kernels_ms_total"))
    o = r.get("other_kernels_ms", {})
    print("   ", sorted(o.items(), key=lambda kv: -kv[1])[:8])
  except Exception as ex:
    print(f, "failed", ex)
PY
