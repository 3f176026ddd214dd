set -u
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/s22
MV_CONV_MAP=1 bash tools/profile_workload.sh r2_greedy_map1 > gpurun_out/s22/prof_greedy.log 2>&1
MV_CONV_MAP=1 bash tools/profile_workload.sh r2_beam_map1 --workload beam > gpurun_out/s22/prof_beam.log 2>&1
ls gpurun_out/prof_r2_greedy_map1 gpurun_out/prof_r2_beam_map1
