set -u
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/s20
(time timeout 1200 python -m pytest tests/test_gpu_edge.py tests/test_gpu_forward.py tests/test_gpu_at_size.py tests/test_gpu_kernels.py -q -x) > gpurun_out/s20/tests.log 2>&1
echo "tests rc $?" >> gpurun_out/s20/tests.log
grep -vE "^\s*$|amdgpu.ids" gpurun_out/s20/tests.log | tail -8
python bench.py --no-cpu-baseline > gpurun_out/s20/bench_greedy.json 2> gpurun_out/s20/bench_greedy.err; tail -1 gpurun_out/s20/bench_greedy.json | cut -c1-300
