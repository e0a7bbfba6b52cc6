R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_run36; mkdir -p $O; cd $R
for v in "A HIFIC_KSPLIT_TARGET_SP128=800" "B HIFIC_KSPLIT_TARGET_SP128=256" "C HIFIC_GC_BIGTILE_MIN_GRID=64" "A HIFIC_KSPLIT_TARGET_SP128=800" "B HIFIC_KSPLIT_TARGET_SP128=256" "C HIFIC_GC_BIGTILE_MIN_GRID=64"; do set -- $v
env $2 HIFIC_BENCH_ROOFLINE_ONLY=1 HIFIC_BENCH_PMC=0 timeout 400 python bench.py --steps 10 --warmup 3 2> $O/err.log | cut -c60-200 | sed "s/^/$1 $2 /" | tee -a $O/ab.log
done
