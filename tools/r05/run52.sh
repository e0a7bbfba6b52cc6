R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_run52; mkdir -p $O; cd $R
for v in 0 1 0 1; do
HIFIC_WGRAD_C3=$v HIFIC_BENCH_ROOFLINE_ONLY=1 HIFIC_BENCH_PMC=0 timeout 400 python bench.py --steps 12 --warmup 3 2> $O/err.log | cut -c60-75,150-180 | sed "s/^/C3=$v /" | tee -a $O/ab.log
done
HIFIC_WRITE_BOUNDS=1 timeout 900 python -m pytest tests/test_gpu_bf16_backward.py -m gpu -q -s -p no:cacheprovider > $O/bounds.log 2>&1; tail -2 $O/bounds.log
cp gpurun_out/bf16_grad_bounds.json $O/ 2>/dev/null
timeout 1200 python -m pytest tests/test_gpu_conv.py tests/test_gpu_golden.py tests/test_gpu_fullsize_backward.py -m gpu -q -x -p no:cacheprovider > $O/tests.log 2>&1; tail -3 $O/tests.log
