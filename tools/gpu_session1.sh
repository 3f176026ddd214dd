#!/bin/bash
# round-2 GPU session 1: parity at the benchmarked sizes + the 2-rank engine test + bench lines
set -u
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/s1
(time timeout 900 python -m pytest tests/test_gpu_at_size.py tests/test_gpu_parallel.py -q -s -x) > gpurun_out/s1/tests.log 2>&1
echo "tests rc $?" >> gpurun_out/s1/tests.log
timeout 300 python bench.py > gpurun_out/s1/bench.json 2> gpurun_out/s1/bench.err
timeout 300 python bench.py --batch 256 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/s1/bench_b256.json 2>> gpurun_out/s1/bench.err
timeout 300 python bench.py --workload beam --steps 3 --warmup 1 > gpurun_out/s1/bench_beam.json 2>> gpurun_out/s1/bench.err
timeout 300 python bench.py --workload train --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/s1/bench_train.json 2>> gpurun_out/s1/bench.err
tail -60 gpurun_out/s1/tests.log
