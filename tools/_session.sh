set -u
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/s15
(time timeout 1200 python -m pytest tests/test_gpu_simaug.py -q -x -s) > gpurun_out/s15/tests.log 2>&1
echo "tests rc $?" >> gpurun_out/s15/tests.log
grep -vE "^\s*$|amdgpu.ids" gpurun_out/s15/tests.log | tail -40
