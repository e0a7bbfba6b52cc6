R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_run5; mkdir -p $O; cd $R
timeout 300 python tools/trace_aten.py > $O/aten.txt 2>&1; grep -A80 "=== ATen" $O/aten.txt | cut -c1-260
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $O/gpu_tests.log 2>&1; tail -5 $O/gpu_tests.log
