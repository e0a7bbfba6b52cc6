R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_run43; mkdir -p $O; cd $R
for n in 1 2 1 2 3; do
HIFIC_SIDE_STREAMS=$n HIFIC_BENCH_ROOFLINE_ONLY=1 HIFIC_BENCH_PMC=0 timeout 400 python bench.py --steps 10 --warmup 3 2> $O/err.log | cut -c60-200 | sed "s/^/SIDE=$n /" | tee -a $O/ab.log
done
HIFIC_SIDE_STREAMS=2 timeout 900 python -m pytest tests/test_gpu_bf16_backward.py tests/test_gpu_modules.py -m gpu -x -q -p no:cacheprovider > $O/tests.log 2>&1; tail -3 $O/tests.log
