#!/bin/bash
# generic GPU session: tools/gpu_session.sh <tag> '<pytest args>' [bench...]
set -u
cd ${GRAFT_REPO_ROOT:-.}
TAG=$1; shift
mkdir -p gpurun_out/$TAG
(time timeout 1500 python -m pytest $1 -q -s) > gpurun_out/$TAG/tests.log 2>&1
echo "tests rc $?" >> gpurun_out/$TAG/tests.log
grep -E "passed|failed|error" gpurun_out/$TAG/tests.log | tail -5
