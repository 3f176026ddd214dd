#!/bin/bash
# GPU sessions, one script: tools/gpu_session_r6.sh <tag> <stage> [<stage> ...]
# (one gpurun call runs the listed stages in order; outputs under gpurun_out/<tag>/).  Stages:
#   margins   the at-size parity tests, the Winograd kernel tests and the reference pins with -s:
#             max|dlogits|, max|dreg|, flips, min oracle margin, per-row beam report
#   suite     the whole GPU suite (-x -q)
#   parallel  tests/test_gpu_parallel.py -s (two ranks on one GPU: gloo, the in-library all-reduce
#             over tests/fake_rccl in both modes, the dropped-event-wait negative controls)
#   wino      tests/test_gpu_wino.py -s (kernel-level parity of the gate kernel forms)
#   bench     the default `python bench.py` line (headline + every sub-workload)
#   headline  `python bench.py --no-sub` x 2 (same-box repeatability of the headline)
#   pmcgreedy rocprofv3 trace + PMC of the greedy workload only
#   profiles  rocprofv3 traces + PMC of greedy / beam / train / bf16 training (tools/profile_workload.sh)
#   smoke     __graft_entry__.smoke()
#   drivercmd the driver's own command line: python bench.py --gpus 1 --steps 20 --warmup 5
#   collect   gather the session's evidence as gpurun_out/<tag>/to_profiles/<tag>_* (copy into profiles/)
#   libab:<name>  headline + beam with build/variants/libmv_<name>.so against the default library
#   ab:<ENV=V>  headline + beam with the env setting against the default, same box
#   trainab:<ENV=V>  training parity tests, then the training step (configs[2]) with the env
#             setting against the default, same box: ms per step and per gate kernel
set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
T=$1; shift
O=gpurun_out/$T
mkdir -p $O
BQ="python bench.py --no-sub --no-cpu-baseline --no-fp32-ref"
line() { python - "$1" <<'PY'
import json, sys
try:
  d = json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
  r = d.get("roofline", {})
  print(sys.argv[1], d.get("value"), d.get("ms_per_step"), "launch ms", r.get("avg_launch_ms"), "frac", r.get("frac"))
except Exception as e:
  print(sys.argv[1], "unreadable:", e)
PY
}
for stage in "$@"; do
  echo "=== stage $stage"
  case $stage in
    margins)
      (time timeout 2400 python -m pytest tests/test_gpu_at_size.py tests/test_gpu_wino.py tests/test_gpu_reference_pin.py tests/test_gpu_trained_parity.py -m gpu -q -s) > $O/margins.log 2>&1
      echo "margins rc $?"; grep -E "passed|failed|error" $O/margins.log | tail -3 ;;
    suite)
      (time timeout 1500 python -m pytest tests -q -m gpu) > $O/gpu_tests.log 2>&1
      echo "suite rc $?"; tail -4 $O/gpu_tests.log ;;
    parallel)
      (time timeout 1200 python -m pytest tests/test_gpu_parallel.py -m gpu -q -s) > $O/parallel_tests.log 2>&1
      echo "parallel rc $?"; grep -E "passed|failed|error|DETECTED|detected|max \|2 ranks" $O/parallel_tests.log | tail -8 ;;
    wino)
      (time timeout 900 python -m pytest tests/test_gpu_wino.py -m gpu -q -s) > $O/wino_tests.log 2>&1
      echo "wino rc $?"; grep -E "passed|failed|error" $O/wino_tests.log | tail -3 ;;
    bench)
      (time timeout 900 python bench.py) > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $?"
      line $O/bench_default.json ;;
    headline)
      for i in 1 2; do timeout 300 $BQ --steps 100 > $O/headline_$i.json 2> $O/headline_$i.err; line $O/headline_$i.json; done ;;
    pmcgreedy)
      bash tools/profile_workload.sh ${T}_greedy > $O/prof_greedy.log 2>&1
      head -6 gpurun_out/prof_${T}_greedy/kernel_trace_stats.md
      python - <<PY
import json, glob
for f in sorted(glob.glob("gpurun_out/prof_${T}_greedy/pmc_convlstm*.json")):
  d = json.load(open(f))
  print(d["kernel"], {k: (round(v, 4) if isinstance(v, float) else v) for k, v in d.items() if k not in ("counters", "kernel", "hbm_bytes_per_launch")},
        {k: round(v / 1e6, 1) for k, v in (d.get("hbm_bytes_per_launch") or {}).items()})
PY
      ;;
    profiles)
      bash tools/profile_workload.sh ${T}_greedy > $O/prof_greedy.log 2>&1
      bash tools/profile_workload.sh ${T}_beam --workload beam > $O/prof_beam.log 2>&1
      bash tools/profile_workload.sh ${T}_train --workload train > $O/prof_train.log 2>&1
      bash tools/profile_workload.sh ${T}_train_bf16 --workload train --batch 64 --compute bf16 --scene-conv-kernel 1 > $O/prof_train_bf16.log 2>&1
      for w in greedy beam train train_bf16; do echo "== $w"; head -8 gpurun_out/prof_${T}_$w/kernel_trace_stats.md; done ;;
    smoke)
      (time timeout 600 python -c "import __graft_entry__ as g; g.smoke()") > $O/smoke.log 2>&1; echo "smoke rc $?"; tail -3 $O/smoke.log ;;
    drivercmd)
      (time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5) > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err; echo "driver cmd rc $?"
      line $O/bench_driver_cmd.json ;;
    collect)
      # copy the session's evidence into profiles/<tag>_* (run LAST; profiles/ travels back only
      # through gpurun_out/, so the copies land in gpurun_out/<tag>/to_profiles/)
      P=$O/to_profiles; mkdir -p $P
      for w in greedy beam train train_bf16; do
        d=gpurun_out/prof_${T}_$w; [ -d $d ] || continue
        cp $d/kernel_trace_stats.md $P/${T}_${w}_kernel_trace_stats.md
        cp $d/bench_under_rocprof.json $P/${T}_${w}_bench_under_rocprof.json
        for f in $d/pmc_*.json; do [ -f $f ] && cp $f $P/${T}_${w}_$(basename $f); done
      done
      [ -f $O/bench_default.json ] && cp $O/bench_default.json $P/${T}_bench_default.json
      [ -f $O/bench_driver_cmd.json ] && cp $O/bench_driver_cmd.json $P/${T}_bench_driver_cmd.json
      [ -f $O/gpu_tests.log ] && tail -5 $O/gpu_tests.log > $P/${T}_gpu_suite_tail.txt
      [ -f $O/smoke.log ] && cp $O/smoke.log $P/${T}_smoke.log
      [ -f $O/margins.log ] && grep -E "rows, max|flips|adversarial|error vs fp64|beam parity|passed|histogram|minADE|minFDE|NLL|worst|trained|rows whose|without an oracle|row-triple vs|configs\[3\]:" $O/margins.log > $P/${T}_parity_margins.log
      ls $P | wc -l ;;
    trainab:*)
      kv=${stage#trainab:}; k=${kv%%=*}
      (time timeout 900 python -m pytest tests/test_gpu_train.py "tests/test_gpu_at_size.py::test_configs2_batch32_train_step_vs_oracle" -m gpu -q -x) > $O/train_tests.log 2>&1
      echo "train tests rc $?"; grep -E "passed|failed|error" $O/train_tests.log | tail -3
      for v in default $k default $k; do
        f=$O/train_$v.json
        if [ $v = default ]; then timeout 300 $BQ --workload train > $f 2> $O/train_$v.err
        else env $kv timeout 300 $BQ --workload train > $f 2> $O/train_$v.err; fi
        python - $f <<'PY'
import json, sys
try:
  d = json.loads(open(sys.argv[1]).read().strip().split("\n")[-1]); r = d["roofline"]
  print(sys.argv[1], d["value"], "traj/s", d["ms_per_step"], "ms/step", r.get("per_kernel_ms"), "frac", r.get("frac"))
except Exception as e:
  print(sys.argv[1], "unreadable:", e)
PY
      done ;;
    ab:*)
      kv=${stage#ab:}; k=${kv%%=*}
      for w in greedy beam; do
        wa=""; [ $w = beam ] && wa="--workload beam --steps 3 --warmup 1"
        [ $w = greedy ] && wa="--steps 100"
        timeout 300 $BQ $wa > $O/ab_${w}_default.json 2> $O/ab_${w}_default.err; line $O/ab_${w}_default.json
        env $kv timeout 300 $BQ $wa > $O/ab_${w}_$k.json 2> $O/ab_${w}_$k.err; line $O/ab_${w}_$k.json
      done ;;
    libab:*)
      name=${stage#libab:}
      for w in greedy beam; do
        wa="--steps 100"; [ $w = beam ] && wa="--workload beam --steps 3 --warmup 1"
        timeout 300 $BQ $wa > $O/libab_${w}_default.json 2> $O/libab_${w}_default.err; line $O/libab_${w}_default.json
        MV_LIB_PATH=$PWD/build/variants/libmv_$name.so timeout 300 $BQ $wa > $O/libab_${w}_$name.json 2> $O/libab_${w}_$name.err; line $O/libab_${w}_$name.json
      done ;;
    *) echo "unknown stage $stage" ;;
  esac
done
