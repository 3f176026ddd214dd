set -u
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/s14
MV_ALLREDUCE=lib-force timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --workload train --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/s14/bench_train_torchrun1.json 2> gpurun_out/s14/bench.err
tail -5 gpurun_out/s14/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/s14/bench_train_torchrun1.json'))
print(d['value'], d['ms_per_step'], d.get('rccl_ranks'), d.get('allreduce'), d['config']['parallelism'])
PY
