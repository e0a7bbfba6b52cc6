R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_run37; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_conv.py -m gpu -x -q -p no:cacheprovider -k "WR_ or weight_resident" > $O/tests.log 2>&1; tail -4 $O/tests.log
