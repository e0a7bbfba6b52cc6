R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_run3; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_conv.py tests/test_gpu_exact_index.py -m gpu -x -q -p no:cacheprovider > $O/tests.log 2>&1; tail -3 $O/tests.log
for pl in 0 1; do
HIFIC_PL=$pl HIFIC_BENCH_ROOFLINE_ONLY=1 HIFIC_BENCH_PMC=0 HIFIC_PROF_DUMP=1 timeout 400 python bench.py --steps 10 --warmup 3 > $O/bench_pl$pl.json 2> $O/dump_pl$pl.log
cut -c1-200 $O/bench_pl$pl.json
done
grep "gconv_pl" $O/dump_pl1.log | awk '{k=$0; sub(/^HIFIC_PROF [^ ]+ [0-9.]+ [0-9.e+]+ /,"",k); n[k]++; t[k]+=$3} END{for(k in n) printf "%6.1f us x%d  %s\n", t[k]/n[k], n[k], k}' | sort -rn
