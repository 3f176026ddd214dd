#!/bin/bash
# Round 4, session s: h2g_q with the whole 1 KB cell row requested before the MFMA chain and the
# packed weights in LDS, against the previous build (build/variants/libmv_head.so, MV_LIB_PATH).
set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r4s
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_forward.py tests/test_gpu_f16x3.py tests/test_gpu_reference_pin.py tests/test_gpu_edge.py -q -x -m gpu > $O/tests.log 2>&1
echo "tests rc $?"; tail -3 $O/tests.log
timeout 600 python -m pytest tests/test_gpu_train_variants.py -q -x -m gpu -k "hidden or emb128" > $O/tests_train.log 2>&1
echo "train variant tests rc $?"; tail -2 $O/tests_train.log
B="python bench.py --no-sub --no-cpu-baseline --no-fp32-ref"
for rep in 1 2; do
  MV_LIB_PATH=build/variants/libmv_head.so timeout 300 $B --steps 80 > $O/greedy_head_$rep.json 2> $O/greedy_head_$rep.err
  timeout 300 $B --steps 80 > $O/greedy_new_$rep.json 2> $O/greedy_new_$rep.err
done
MV_LIB_PATH=build/variants/libmv_head.so timeout 300 $B --workload beam > $O/beam_head.json 2> $O/beam_head.err
timeout 300 $B --workload beam > $O/beam_new.json 2> $O/beam_new.err
MV_LIB_PATH=build/variants/libmv_head.so timeout 300 $B --workload train > $O/train_head.json 2> $O/train_head.err
timeout 300 $B --workload train > $O/train_new.json 2> $O/train_new.err
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/*.json")):
  try:
    d = json.load(open(f)); r = d["roofline"]
    o = r.get("other_kernels_ms", {})
    print(f.split("/")[-1], d["value"], d["ms_per_step"], r.get("avg_launch_ms"), r.get("other_kernels_ms_total"),
          "h2g", o.get("hidden2grid"), "gnn", o.get("gnn_attend"), "tail", o.get("decode_tail"))
  except Exception as ex:
    print(f, "failed", ex)
PY
