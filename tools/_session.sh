set -u
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/s28
mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_forward.py tests/test_gpu_edge.py -q -x) > $O/tests.log 2>&1
echo "tests rc $?" >> $O/tests.log
grep -E "passed|failed|error|rc " $O/tests.log | tail -3
python bench.py > $O/bench_greedy.json 2> $O/bench_greedy.err
python bench.py --batch 256 --no-cpu-baseline > $O/bench_greedy_b256.json 2>/dev/null
python bench.py --workload beam --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_beam.json 2>/dev/null
python bench.py --workload train > $O/bench_train.json 2>/dev/null
for w in greedy beam train; do
  python bench.py --workload $w --compute bf16 --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_${w}_bf16.json 2>/dev/null
done
for f in $O/bench_*.json; do python - "$f" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline']
print(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], 'frac', r['frac'], 'launch', r['avg_launch_ms'], 'other', r['other_kernels_ms_total'], 'cpu', (d.get('cpu_baseline') or {}).get('value'))
PY
done
bash tools/profile_workload.sh r2_greedy > $O/prof_greedy.log 2>&1
bash tools/profile_workload.sh r2_beam --workload beam > $O/prof_beam.log 2>&1
bash tools/profile_workload.sh r2_train --workload train > $O/prof_train.log 2>&1
ls gpurun_out/prof_r2_greedy | head -30
