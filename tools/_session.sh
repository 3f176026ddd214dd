set -u
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/s10
(time timeout 1200 python -m pytest tests/test_gpu_train_variants.py tests/test_gpu_train.py tests/test_gpu_kernels.py -q -x -s) > gpurun_out/s10/tests.log 2>&1
echo "tests rc $?" >> gpurun_out/s10/tests.log
grep -E "^[a-z0-9_]+/f|step [0-9]:|worst|teacher forcing|passed|failed|Error|error" gpurun_out/s10/tests.log | tail -60
timeout 300 python bench.py --workload beam --steps 3 --warmup 1 --no-cpu-baseline > "gpurun_out/s10/bench_beam.json" 2>> gpurun_out/s10/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/s10/bench_beam.json')); r=d['roofline']
print('beam', d['value'], d['ms_per_step'], r['avg_launch_ms'], r['other_kernels_ms'])
PY
