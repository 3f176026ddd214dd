cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6s
(timeout 900 python -m pytest tests/test_gpu_train.py -q -s -k "f16x3" ) > gpurun_out/r6s/wino_tests.log 2>&1; echo rc $?
(MV_WGRAD_WINO=0 timeout 900 python -m pytest tests/test_gpu_train.py -q -s -k "f16x3_gradients_match" ) > gpurun_out/r6s/direct_tests.log 2>&1; echo rc $?
grep -E "h rows|x rows|passed|failed" gpurun_out/r6s/wino_tests.log | awk '{print $1,$2,$3,$4,$5}' | sort -k5 -g | tail -8
echo ---- direct
grep -E "h rows|x rows|passed|failed" gpurun_out/r6s/direct_tests.log | awk '{print $1,$2,$3,$4,$5}' | sort -k5 -g | tail -6
for v in 1 0; do MV_WGRAD_WINO=$v timeout 600 python bench.py --only-sub train_n32 --no-cpu-baseline --no-fp32-ref --steps 5 > gpurun_out/r6s/bench_wino$v.json 2> gpurun_out/r6s/bench_wino$v.err; python - <<PY
import json
d=json.loads(open('gpurun_out/r6s/bench_wino$v.json').read().strip().split('\n')[-1])
t=d['train_n32']; print('wino$v', t['value'], t['ms_per_step'], t['roofline']['per_kernel_ms'], t['roofline'].get('other_kernels_ms_total'))
PY
done
