R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_run32; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_two_ranks.py -q -x -p no:cacheprovider > $O/t.log 2>&1; tail -5 $O/t.log
