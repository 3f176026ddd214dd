set -u
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/s27
(time timeout 1500 python -m pytest tests/test_gpu_train.py tests/test_gpu_train_variants.py tests/test_gpu_reference_pin.py tests/test_gpu_at_size.py tests/test_gpu_kernels.py -q -x) > gpurun_out/s27/tests.log 2>&1
echo "tests rc $?" >> gpurun_out/s27/tests.log
grep -E "passed|failed|error|rc " gpurun_out/s27/tests.log | tail -5
python bench.py --workload train --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/s27/bench_train.json
python - <<'PY'
import json
d=json.loads(open('gpurun_out/s27/bench_train.json').read()); r=d['roofline']
print('train', d['value'], d['ms_per_step'], r.get('per_kernel_ms'), r['other_kernels_ms_total'])
print({k:v for k,v in sorted(r['other_kernels_ms'].items(), key=lambda kv:-kv[1])[:14]})
PY
python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('greedy', d['value'], r['avg_launch_ms'], r['other_kernels_ms_total'])"
