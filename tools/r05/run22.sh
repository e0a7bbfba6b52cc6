R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_run22; mkdir -p $O; cd $R
HIFIC_WRITE_BOUNDS=1 timeout 1500 python -m pytest tests/test_gpu_golden.py tests/test_gpu_bf16_backward.py -m gpu -q -s -p no:cacheprovider > $O/tests.log 2>&1; tail -3 $O/tests.log
grep -h "worst grad-norm rel\|reconstruction given equal\|measured table" $O/tests.log | cut -c1-400
cp gpurun_out/bf16_grad_bounds.json $O/ 2>/dev/null
