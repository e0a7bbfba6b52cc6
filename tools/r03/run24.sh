# new defaults (four-quarter sp9 form, 16-byte ChannelNorm kernels): the test files they can affect, before the final pass
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_run24; mkdir -p $O; cd $R
timeout 500 python -m pytest tests/test_gpu_exact_index.py tests/test_gpu_golden.py tests/test_gpu_modules.py tests/test_gpu_elementwise.py tests/test_gpu_conv.py -q -p no:cacheprovider > $O/tests.log 2>&1
tail -2 $O/tests.log; grep -E "^FAILED|^ERROR" $O/tests.log | cut -c1-250
