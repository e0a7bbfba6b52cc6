R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_run24; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_modules.py -m gpu -x -q -s -p no:cacheprovider -k "context_gradient or discriminator" > $O/tests.log 2>&1; tail -4 $O/tests.log; grep "context gradient" $O/tests.log
for d1 in 0 1; do
HIFIC_D1_FUSED=$d1 HIFIC_BENCH_ROOFLINE_ONLY=1 HIFIC_BENCH_PMC=0 timeout 400 python bench.py --steps 10 --warmup 3 > $O/bench_d1$d1.json 2> $O/err_d1$d1.log
cut -c1-200 $O/bench_d1$d1.json
done
