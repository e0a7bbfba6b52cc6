# round-6 first call: where the host time and the ATen launches of the round-5 build are (inputs of items 2 and 3)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_run1; mkdir -p $O; cd $R
timeout 300 python bench.py --steps 10 --warmup 3 --no-extras > $O/bench_base.json 2> $O/bench_base.err; cut -c1-400 $O/bench_base.json
timeout 300 python tools/host_floor_probe.py profile > $O/hostprof.txt 2>&1; head -60 $O/hostprof.txt
timeout 300 python tools/trace_aten.py > $O/aten.txt 2>&1; tail -40 $O/aten.txt
echo done
