R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_run44; mkdir -p $O; cd $R
for n in 1 3 2 1 3 2 1 3 2; do
HIFIC_SIDE_STREAMS=$n HIFIC_BENCH_ROOFLINE_ONLY=1 HIFIC_BENCH_PMC=0 timeout 400 python bench.py --steps 12 --warmup 3 2> $O/err.log | cut -c60-75,150-180 | sed "s/^/SIDE=$n /" | tee -a $O/ab.log
done
