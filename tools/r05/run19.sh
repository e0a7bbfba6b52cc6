R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_run19; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_conv.py -m gpu -x -q -p no:cacheprovider -k "WR_ or weight_resident" > $O/tests.log 2>&1; tail -5 $O/tests.log
for wr in 0 1; do
for cfg in "16 9 60 256 7 1 3,3,3,3 fwd" "16 60 3 256 7 1 3,3,3,3 fwd" "16 60 3 256 7 1 3,3,3,3 bwd" "32 15 64 256 4 2 1,1,1,1 fwd"; do
  set -- $cfg
  echo -n "WR=$wr C$2 K$3 H$4 R$5 s$6 $8: " >> $O/ab.log
  HIFIC_WR=$wr MPROF=1 MN=$1 MC=$2 MK=$3 MH=$4 MR=$5 MS=$6 MPAD=$7 timeout 120 python tools/micro_conv.py $8 20 2>&1 | grep -E "^(fwd|bwd)" | sed 's/^[^[]*\[/[/' >> $O/ab.log
done; done
for ws in 0 2; do
  echo -n "WR=1 WSTAGE=$ws C60 K3 fwd: " >> $O/ab.log
  HIFIC_WR_WSTAGE=$ws MPROF=1 MN=16 MC=60 MK=3 MH=256 MR=7 MS=1 timeout 120 python tools/micro_conv.py fwd 20 2>&1 | grep -E "^fwd" | sed 's/^[^[]*\[/[/' >> $O/ab.log
done
cat $O/ab.log
