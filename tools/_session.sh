set -u
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/s34
mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_simaug.py -q -x -s -k "mixup or experiment_3") > $O/tests.log 2>&1
echo "tests rc $?" >> $O/tests.log
grep -vE "^\s*$|amdgpu.ids" $O/tests.log | tail -25
