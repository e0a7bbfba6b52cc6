R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_run9; mkdir -p $O; cd $R
for ag in 0 1 2 3; do
HIFIC_BENCH_ROOFLINE_ONLY=1 HIFIC_SP9_AG=$ag HIFIC_PROF_DUMP=1 timeout 300 python bench.py --steps 4 --warmup 2 > $O/b_ag$ag.json 2> $O/prof_ag$ag.log
echo "AG=$ag $(python -c "import json; d=json.load(open('$O/b_ag$ag.json')); print(d['ms_per_step'], {k: round(v['avg_launch_us'],1) for k,v in d['roofline']['per_kernel'].items() if 'sp9_kernel<2,2' in k})")"
grep "sp9_kernel<2,2" $O/prof_ag$ag.log | grep "K960 C960" | awk '{n[$2]++; s[$2]+=$3} END {for (k in n) print "   ", k, n[k], s[k]/n[k]}'
done
