R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_run28; mkdir -p $O; cd $R
for d1 in 0 1 0 1; do
HIFIC_D1_FUSED=$d1 HIFIC_BENCH_ROOFLINE_ONLY=1 HIFIC_BENCH_PMC=0 timeout 400 python bench.py --steps 10 --warmup 3 2> $O/err.log | cut -c1-180 | sed "s/^/D1=$d1 /" | tee -a $O/ab.log
done
