R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_run30; mkdir -p $O; cd $R
HIFIC_WRITE_BOUNDS=1 timeout 900 python -m pytest tests/test_gpu_bf16_backward.py -m gpu -q -s -p no:cacheprovider > $O/tests.log 2>&1; tail -3 $O/tests.log
cp gpurun_out/bf16_grad_bounds.json $O/ 2>/dev/null
timeout 900 python -m pytest tests/test_gpu_golden.py -m gpu -q -s -p no:cacheprovider > $O/golden.log 2>&1; tail -2 $O/golden.log
grep -h "worst grad-norm rel" $O/golden.log | sed 's/.*\] \([a-z_A-Z0-9]*\):.*worst grad-norm rel \(.*\)/\1 \2/'
