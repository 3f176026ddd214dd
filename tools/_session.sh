set -u
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/s31
mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_forward.py tests/test_gpu_edge.py -q -x) > $O/tests.log 2>&1
echo "tests rc $?" >> $O/tests.log
grep -E "passed|failed|error|rc " $O/tests.log | tail -3
for lib in libmultiverse_hip.so libmv_gnn32.so libmultiverse_hip.so libmv_gnn32.so; do
MV_LIB_PATH=$PWD/multiverse_amd/$lib python bench.py --workload beam --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('beam $lib', d['value'], r['other_kernels_ms']['gnn_attend'], r['other_kernels_ms_total'])"
done
for lib in libmultiverse_hip.so libmv_gnn32.so; do
MV_LIB_PATH=$PWD/multiverse_amd/$lib python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('greedy $lib', d['value'], r['other_kernels_ms']['gnn_attend'], r['other_kernels_ms_total'])"
done
