set -u
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/s26
(time timeout 1500 python -m pytest tests/test_gpu_parallel.py tests/test_gpu_reference_pin.py tests/test_gpu_simaug.py tests/test_gpu_train.py tests/test_gpu_train_variants.py tests/test_gpu_cli.py tests/test_gpu_dropin.py -q -x) > gpurun_out/s26/tests.log 2>&1
echo "tests rc $?" >> gpurun_out/s26/tests.log
grep -E "passed|failed|error|rc " gpurun_out/s26/tests.log | tail -5
