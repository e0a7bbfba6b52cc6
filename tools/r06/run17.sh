R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_run17; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_exact_generator.py tests/test_gpu_exact_index.py tests/test_gpu_golden.py -q -x -p no:cacheprovider > $O/t.log 2>&1; tail -4 $O/t.log
timeout 300 python tools/r06/exact_ab.py 8 > $O/exact_ab.log 2>&1; grep exact_ab $O/exact_ab.log
HIFIC_SPLIT_IN_PACK=0 timeout 300 python tools/r06/exact_ab.py 8 > $O/exact_ab0.log 2>&1; grep exact_ab $O/exact_ab0.log | sed 's/^/[SPLIT_IN_PACK=0] /'
