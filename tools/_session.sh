set -u
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/s11
(time timeout 1200 python -m pytest tests/test_gpu_bf16.py -q -x -s) > gpurun_out/s11/tests.log 2>&1
echo "tests rc $?" >> gpurun_out/s11/tests.log
grep -E "bf16|f16x3|largest|worst|passed|failed|Error|error|assert" gpurun_out/s11/tests.log | tail -40
timeout 300 python bench.py --compute bf16 --no-cpu-baseline --steps 10 --warmup 3 > "gpurun_out/s11/bench_greedy_bf16.json" 2>> gpurun_out/s11/bench.err
timeout 300 python bench.py --compute bf16 --workload beam --steps 3 --warmup 1 --no-cpu-baseline > "gpurun_out/s11/bench_beam_bf16.json" 2>> gpurun_out/s11/bench.err
timeout 300 python bench.py --compute bf16 --workload train --steps 5 --warmup 2 --no-cpu-baseline > "gpurun_out/s11/bench_train_bf16.json" 2>> gpurun_out/s11/bench.err
tail -3 gpurun_out/s11/bench.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/s11/bench_*.json')):
  try:
    d=json.load(open(f)); r=d['roofline']
    print(f.split('/')[-1], d['value'], d['ms_per_step'], 'conv ms', r['avg_launch_ms'], 'frac', r['frac'], r.get('per_kernel_ms'), 'other', r['other_kernels_ms_total'])
  except Exception as e: print(f, 'ERR', e)
PY
