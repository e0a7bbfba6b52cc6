R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_run15; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_tickets.py tests/test_gpu_modules.py -q -x -p no:cacheprovider > $O/t.log 2>&1; tail -3 $O/t.log
