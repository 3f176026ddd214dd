cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6t
for sp in 0 4 8 12 17 21; do
  if [ $sp = 0 ]; then E=""; else E="MV_WGRAD_WIDE_SPLITS=$sp"; fi
  env $E timeout 600 python bench.py --workload train --no-sub --no-cpu-baseline --no-fp32-ref --steps 20 > gpurun_out/r6t/bench_sp$sp.json 2> gpurun_out/r6t/bench_sp$sp.err
  python - <<PY
import json
d=json.loads(open('gpurun_out/r6t/bench_sp$sp.json').read().strip().split('\n')[-1])
r=d['roofline']; print('splits $sp', d['value'], d['ms_per_step'], r.get('per_kernel_ms'), r.get('other_kernels_ms_total'))
PY
done
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r6t
rocprofv3 --kernel-trace --stats -d $O -o kt -- python $GRAFT_REPO_ROOT/bench.py --workload train --no-sub --no-cpu-baseline --no-fp32-ref --steps 5 --warmup 1 > $O/under_rocprof.json 2> $O/kt.err
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $O/kt_results.db > $O/kernel_trace_stats.md 2>> $O/kt.err
rm -f $O/*.db
head -30 $O/kernel_trace_stats.md | cut -d'|' -f2-8
