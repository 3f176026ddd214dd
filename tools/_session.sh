set -u
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/s37
mkdir -p $O
(time timeout 1700 python -m pytest tests/test_gpu_parallel.py tests/test_gpu_reference_pin.py tests/test_gpu_simaug.py tests/test_gpu_train.py tests/test_gpu_train_variants.py -q -x) > $O/tests.log 2>&1
echo "tests rc $?" >> $O/tests.log
grep -E "passed|failed|error|rc |real" $O/tests.log | tail -5
