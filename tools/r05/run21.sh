R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_run21; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_conv.py -m gpu -x -q -p no:cacheprovider -k "WR_ or weight_resident or G9 or pipelined" > $O/tests.log 2>&1; tail -3 $O/tests.log
for lw in 0 1; do
  echo -n "LW8=$lw C60 K3 fwd: " >> $O/ab.log
  HIFIC_WR_LW8=$lw MPROF=1 MN=16 MC=60 MK=3 MH=256 MR=7 MS=1 timeout 120 python tools/micro_conv.py fwd 20 2>&1 | grep -E "^fwd" | sed 's/^[^[]*\[/[/' >> $O/ab.log
done
cat $O/ab.log
for wr in 0 1; do
HIFIC_WR=$wr HIFIC_BENCH_ROOFLINE_ONLY=1 HIFIC_BENCH_PMC=0 HIFIC_PROF_DUMP=1 timeout 400 python bench.py --steps 10 --warmup 3 > $O/bench_wr$wr.json 2> $O/dump_wr$wr.log
cut -c1-200 $O/bench_wr$wr.json
grep -E "gconv_wr|K60 C9|K21 C60|K64 C15" $O/dump_wr$wr.log | awk '{k=$0; sub(/^HIFIC_PROF [^ ]+ [0-9.]+ [0-9.e+]+ /,"",k); n[k]++; t[k]+=$3} END{for(k in n) printf "%6.1f us x%d  %s\n", t[k]/n[k], n[k], k}' | sort -rn
done
