# per-launch dump of the headline cycle (in-library profiler, single stream) + baseline headline
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_run1; mkdir -p $O; cd $R
HIFIC_BENCH_ROOFLINE_ONLY=1 HIFIC_BENCH_PMC=0 HIFIC_PROF_DUMP=1 timeout 400 python bench.py --steps 10 --warmup 3 > $O/bench.json 2> $O/dump.log
cut -c1-400 $O/bench.json
python tools/prof_layers.py $O/dump.log 1 > $O/layers.md; head -5 $O/layers.md
