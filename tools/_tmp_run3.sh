cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6u
for v in 1 0; do
(MV_WGRAD_WINO_BF16=$v timeout 900 python -m pytest tests/test_gpu_bf16.py -q -s ) > gpurun_out/r6u/bf16_tests_wino$v.log 2>&1; echo rc $?
grep -E "cosine|passed|failed" gpurun_out/r6u/bf16_tests_wino$v.log
done
for v in 1 0 1 0; do MV_WGRAD_WINO_BF16=$v timeout 600 python bench.py --only-sub train_bf16_n64 --no-cpu-baseline --no-fp32-ref --steps 5 > gpurun_out/r6u/bench_wino$v.json 2> gpurun_out/r6u/bench_wino$v.err; python - <<PY
import json
d=json.loads(open('gpurun_out/r6u/bench_wino$v.json').read().strip().split('\n')[-1])
t=d['train_bf16_n64']; print('wino$v', t['value'], t['ms_per_step'], t['roofline']['per_kernel_ms'], t['roofline'].get('other_kernels_ms_total'))
PY
done
