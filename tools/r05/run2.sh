# new pipelined stride-2 forward kernel: parity tests + A/B micro timings (HIFIC_PL=0/1), kernel alone via the in-library profiler
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_run2; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_conv.py tests/test_gpu_exact_index.py -m gpu -x -q -p no:cacheprovider > $O/tests.log 2>&1; tail -5 $O/tests.log
for cfg in "16 60 120 256" "16 120 240 128" "16 240 480 64" "16 480 960 32"; do
  set -- $cfg
  for pl in 0 1; do
    echo "== N$1 C$2 K$3 H$4 PL=$pl" >> $O/micro.log
    HIFIC_PL=$pl MPROF=1 MN=$1 MC=$2 MK=$3 MH=$4 MR=3 MS=2 timeout 120 python tools/micro_conv.py fwd 20 >> $O/micro.log 2>&1
  done
done
for cfg in "32 64 128 128" "32 128 256 64" "32 256 512 32"; do
  set -- $cfg
  for pl in 0 1; do
    echo "== 4x4 N$1 C$2 K$3 H$4 PL=$pl" >> $O/micro.log
    HIFIC_PL=$pl MPROF=1 MN=$1 MC=$2 MK=$3 MH=$4 MR=4 MS=2 MPAD=1,1,1,1 timeout 120 python tools/micro_conv.py fwd 20 >> $O/micro.log 2>&1
  done
done
cat $O/micro.log
