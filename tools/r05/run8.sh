R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_run8; mkdir -p $O; cd $R
timeout 800 python -m pytest tests/test_gpu_golden.py -m gpu -x -q -s -p no:cacheprovider -k "exact_training or exact_reconstruction" > $O/tests.log 2>&1; tail -12 $O/tests.log | cut -c1-400
HIFIC_EXACT_TRAIN=1 timeout 300 python bench.py --steps 6 --warmup 3 --no-extras > $O/bench_exact.json 2> $O/bench_exact.err; cut -c1-250 $O/bench_exact.json; tail -3 $O/bench_exact.err
