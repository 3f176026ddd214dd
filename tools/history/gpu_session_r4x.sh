#!/bin/bash
# Round 4, session x: the LSTM update over common denominators in the inference epilogues
# (lstm_update_fused: 5 v_exp + 2 v_rcp per element instead of 5 + 5) against the previous build
# (build/variants/libmv_head.so, MV_LIB_PATH).
# (lstm_update_fused was removed after this session: zero gain, profiles/r4x_fused_lstm_update.txt.)
set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r4x
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_wino.py tests/test_gpu_forward.py tests/test_gpu_f16x3.py tests/test_gpu_at_size.py tests/test_gpu_reference_pin.py tests/test_gpu_bf16.py -q -x -m gpu > $O/tests.log 2>&1
echo "tests rc $?"; tail -3 $O/tests.log
B="python bench.py --no-sub --no-cpu-baseline --no-fp32-ref"
for rep in 1 2; do
  MV_LIB_PATH=build/variants/libmv_head.so timeout 300 $B --steps 80 > $O/greedy_head_$rep.json 2> $O/greedy_head_$rep.err
  timeout 300 $B --steps 80 > $O/greedy_new_$rep.json 2> $O/greedy_new_$rep.err
done
MV_LIB_PATH=build/variants/libmv_head.so timeout 300 $B --workload beam > $O/beam_head.json 2> $O/beam_head.err
timeout 300 $B --workload beam > $O/beam_new.json 2> $O/beam_new.err
MV_LIB_PATH=build/variants/libmv_head.so timeout 300 $B --compute bf16 --scene-conv-kernel 1 --steps 80 > $O/bf16_head.json 2> $O/bf16_head.err
timeout 300 $B --compute bf16 --scene-conv-kernel 1 --steps 80 > $O/bf16_new.json 2> $O/bf16_new.err
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/*.json")):
  try:
    d = json.load(open(f)); r = d["roofline"]
    print(f.split("/")[-1], d["value"], d["ms_per_step"], r.get("avg_launch_ms"), r.get("frac"))
  except Exception as ex:
    print(f, "failed", ex)
PY
