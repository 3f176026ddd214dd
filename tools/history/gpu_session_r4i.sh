#!/bin/bash
# Round 4, session i: hidden sizes 128 / 512; full suite.
set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r4i
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_edge.py -q -x -s -k "hidden_size or rejects" > $O/hidden.log 2>&1
echo "hidden tests rc $?"; grep -E "hidden|passed|failed|Error|error" $O/hidden.log | tail -24
timeout 1200 python -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1; rc=$?
echo "gpu tests rc $rc"; tail -5 $O/gpu_tests.log
if [ $rc -ne 0 ]; then grep -E "^(FAILED|ERROR)" $O/gpu_tests.log | head -40; fi
