R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_run18; mkdir -p $O; cd $R
timeout 300 python -m pytest tests/test_gpu_conv.py tests/test_gpu_exact_index.py -m gpu -x -q -p no:cacheprovider > $O/tests.log 2>&1; tail -3 $O/tests.log
for ws in 0 1; do
for cfg in "16 480 960 32" "16 60 120 256" "16 240 480 64"; do
  set -- $cfg
  echo -n "WS=$ws C$2 K$3 H$4: " >> $O/ws.log
  HIFIC_PL_WS=$ws MPROF=1 MN=$1 MC=$2 MK=$3 MH=$4 MR=3 MS=2 timeout 120 python tools/micro_conv.py fwd 20 2>&1 | grep "^fwd" | sed 's/.*\[//' >> $O/ws.log
done; done
cat $O/ws.log
for ws in 0 1; do
HIFIC_PL_WS=$ws HIFIC_BENCH_ROOFLINE_ONLY=1 HIFIC_BENCH_PMC=0 HIFIC_PROF_DUMP=1 timeout 400 python bench.py --steps 10 --warmup 3 > $O/bench_ws$ws.json 2> $O/dump_ws$ws.log
cut -c1-200 $O/bench_ws$ws.json
grep "gconv_pl" $O/dump_ws$ws.log | awk '{k=$0; sub(/^HIFIC_PROF [^ ]+ [0-9.]+ [0-9.e+]+ /,"",k); n[k]++; t[k]+=$3} END{for(k in n) printf "%6.1f us x%d  %s\n", t[k]/n[k], n[k], k}' | sort -rn
done
