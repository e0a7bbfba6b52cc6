R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${OUT:-r05_run13}; mkdir -p $O; cd $R
for v in 1 0; do
HIFIC_PLT=$v HIFIC_BENCH_ROOFLINE_ONLY=1 HIFIC_BENCH_PMC=0 HIFIC_PROF_DUMP=1 timeout 400 python bench.py --steps 10 --warmup 3 > $O/bench_plt$v.json 2> $O/dump_plt$v.log
cut -c1-200 $O/bench_plt$v.json
done
grep "gconv_plt\|gconv_mp\|phs" $O/dump_plt1.log | awk '{k=$0; sub(/^HIFIC_PROF [^ ]+( narrow)? [0-9.]+ [0-9.e+]+ /,"",k); n[k]++; t[k]+=($3 ~ /^[0-9.]+$/ ? $3 : $4)} END{for(k in n) printf "%6.1f us x%d  %s\n", t[k]/n[k], n[k], k}' | sort -rn | head -20
echo ---- PLT=0
grep "gconv_plt\|gconv_mp\|phs" $O/dump_plt0.log | awk '{k=$0; sub(/^HIFIC_PROF [^ ]+( narrow)? [0-9.]+ [0-9.e+]+ /,"",k); n[k]++; t[k]+=($3 ~ /^[0-9.]+$/ ? $3 : $4)} END{for(k in n) printf "%6.1f us x%d  %s\n", t[k]/n[k], n[k], k}' | sort -rn | head -20
