set -u
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/s19
(time timeout 1200 python -m pytest tests/test_gpu_edge.py tests/test_gpu_forward.py tests/test_gpu_at_size.py tests/test_gpu_kernels.py tests/test_gpu_f16x3.py -q -x -k "beam or shared or graph") > gpurun_out/s19/tests.log 2>&1
echo "tests rc $?" >> gpurun_out/s19/tests.log
grep -vE "^\s*$|amdgpu.ids" gpurun_out/s19/tests.log | tail -15
python bench.py --workload beam --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/s19/bench_beam.json 2> gpurun_out/s19/bench_beam.err; tail -1 gpurun_out/s19/bench_beam.json | cut -c1-420
