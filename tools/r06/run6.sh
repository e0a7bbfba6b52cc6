R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_run6; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_tickets.py tests/test_gpu_modules.py tests/test_gpu_elementwise.py -q -x -p no:cacheprovider > $O/t1.log 2>&1; tail -25 $O/t1.log
timeout 300 python bench.py --steps 10 --warmup 3 --no-extras > $O/bench_tk1.json 2>$O/bench_tk1.err; cut -c1-200 $O/bench_tk1.json
HIFIC_TICKETS=0 timeout 300 python bench.py --steps 10 --warmup 3 --no-extras > $O/bench_tk0.json 2>$O/bench_tk0.err; cut -c1-200 $O/bench_tk0.json
timeout 300 python bench.py --steps 10 --warmup 3 --no-extras > $O/bench_tk1b.json 2>$O/bench_tk1b.err; cut -c1-200 $O/bench_tk1b.json
