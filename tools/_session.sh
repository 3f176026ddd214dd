set -u
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/s35
mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_edge.py tests/test_gpu_forward.py tests/test_gpu_reference_pin.py -q -x -k "beam or shared") > $O/tests.log 2>&1
echo "tests rc $?" >> $O/tests.log
grep -E "passed|failed|error|rc " $O/tests.log | tail -3
for m in 1 0 1 0; do
MV_BEAM_GNN_DEDUPE=$m python bench.py --workload beam --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('beam dedupe $m', d['value'], r['other_kernels_ms']['gnn_attend'], r['other_kernels_ms_total'])"
done
