R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_run6; mkdir -p $O; cd $R
P=$R/high-fidelity-generative-compression_amd
timeout 600 python -m pytest tests/test_gpu_conv.py tests/test_gpu_exact_index.py -m gpu -x -q -p no:cacheprovider > $O/tests.log 2>&1; tail -3 $O/tests.log
for cfg in "16 60 120 256" "16 480 960 32"; do
  set -- $cfg
  for abl in 0 1 2 4 8 16 64 128; do
    lib=$P/libhific_hip_abl$abl.so; [ $abl = 0 ] && lib=$P/libhific_hip.so
    echo -n "C$2 K$3 H$4 abl=$abl: " >> $O/abl.log
    HIFIC_LIB_PATH=$lib MPROF=1 MN=$1 MC=$2 MK=$3 MH=$4 MR=3 MS=2 timeout 120 python tools/micro_conv.py fwd 20 2>&1 | grep "^fwd" >> $O/abl.log
  done
done
cat $O/abl.log
HIFIC_BENCH_ROOFLINE_ONLY=1 HIFIC_BENCH_PMC=0 HIFIC_PROF_DUMP=1 timeout 400 python bench.py --steps 10 --warmup 3 > $O/bench.json 2> $O/dump.log
cut -c1-200 $O/bench.json
grep "gconv_pl" $O/dump.log | awk '{k=$0; sub(/^HIFIC_PROF [^ ]+ [0-9.]+ [0-9.e+]+ /,"",k); n[k]++; t[k]+=$3} END{for(k in n) printf "%6.1f us x%d  %s\n", t[k]/n[k], n[k], k}' | sort -rn
