R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_run12; mkdir -p $O; cd $R
HIFIC_EXACT_TRAIN=1 HIFIC_BENCH_ROOFLINE_ONLY=1 HIFIC_BENCH_PMC=0 HIFIC_PROF_DUMP=1 timeout 400 python bench.py --steps 6 --warmup 3 > $O/bench.json 2> $O/dump.log
cut -c1-200 $O/bench.json
