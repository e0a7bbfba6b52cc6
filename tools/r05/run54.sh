R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_run54; mkdir -p $O; cd $R
for v in 0 1 0 1 0 1; do
HIFIC_EXP_SKIP_DGEN=$v HIFIC_BENCH_ROOFLINE_ONLY=1 HIFIC_BENCH_PMC=0 timeout 400 python bench.py --steps 12 --warmup 3 2> $O/err.log | cut -c60-75,150-180 | sed "s/^/SKIPDGEN=$v /" | tee -a $O/ab.log
done
