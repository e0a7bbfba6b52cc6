R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${OUT:-r05_run7}; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_conv.py tests/test_gpu_exact_index.py -m gpu -x -q -p no:cacheprovider > $O/tests.log 2>&1; tail -3 $O/tests.log
for cfg in "16 60 120 256" "16 480 960 32"; do
  set -- $cfg
  echo -n "C$2 K$3 H$4: " >> $O/micro.log
  MPROF=1 MN=$1 MC=$2 MK=$3 MH=$4 MR=3 MS=2 timeout 120 python tools/micro_conv.py fwd 20 2>&1 | grep "^fwd" >> $O/micro.log
done
cat $O/micro.log
HIFIC_BENCH_ROOFLINE_ONLY=1 HIFIC_BENCH_PMC=0 HIFIC_PROF_DUMP=1 timeout 400 python bench.py --steps 10 --warmup 3 > $O/bench.json 2> $O/dump.log
cut -c1-200 $O/bench.json
grep "gconv_pl" $O/dump.log | awk '{k=$0; sub(/^HIFIC_PROF [^ ]+ [0-9.]+ [0-9.e+]+ /,"",k); n[k]++; t[k]+=$3} END{for(k in n) printf "%6.1f us x%d  %s\n", t[k]/n[k], n[k], k}' | sort -rn
