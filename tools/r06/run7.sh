R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_run7; mkdir -p $O; cd $R
HIFIC_BENCH_DIAG=1 timeout 300 python bench.py --steps 10 --warmup 3 --no-extras > $O/bench.json 2>$O/bench.err; cut -c1-200 $O/bench.json; grep -o '"launch_modes.*' $O/bench.json | cut -c1-600
timeout 300 python tools/trace_aten.py > $O/aten.txt 2>&1; grep -A40 "=== ATen" $O/aten.txt | cut -c1-200
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $O/gpu_tests.log 2>&1; tail -5 $O/gpu_tests.log
