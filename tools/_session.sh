set -u
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/s8
for v in w4 w8; do
  export MV_LIB_PATH=$PWD/multiverse_amd/libmv_conv_$v.so
  (timeout 600 python -m pytest tests/test_gpu_f16x3.py tests/test_gpu_train.py -q -x -k f16x3 2>&1 | tail -3) > gpurun_out/s8/tests_$v.log 2>&1
  echo "$v tests: $(tail -1 gpurun_out/s8/tests_$v.log)"
  for rep in 1 2; do
  timeout 300 python bench.py --no-cpu-baseline --steps 10 --warmup 3 > "gpurun_out/s8/bench_greedy_${v}_$rep.json" 2>> gpurun_out/s8/bench.err
  done
  timeout 300 python bench.py --workload train --steps 5 --warmup 2 --no-cpu-baseline > "gpurun_out/s8/bench_train_$v.json" 2>> gpurun_out/s8/bench.err
  timeout 300 python bench.py --workload beam --steps 3 --warmup 1 --no-cpu-baseline > "gpurun_out/s8/bench_beam_$v.json" 2>> gpurun_out/s8/bench.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/s8/bench_*.json')):
  try:
    d=json.load(open(f)); r=d['roofline']
    print(f.split('/')[-1], d['value'], d['ms_per_step'], 'conv ms', r['avg_launch_ms'], r.get('per_kernel_ms'), 'other', r['other_kernels_ms_total'])
  except Exception as e: print(f, 'ERR', e)
PY
