set -u
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/s38
mkdir -p $O
(time timeout 1200 python -m pytest tests/test_gpu_train.py tests/test_gpu_train_variants.py tests/test_gpu_simaug.py -q -x -k "gradients_match_oracle or loss_and_decoder or single or mixup") > $O/tests.log 2>&1
echo "tests rc $?" >> $O/tests.log
grep -E "passed|failed|error|rc |real" $O/tests.log | tail -5
