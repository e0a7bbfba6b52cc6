R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_run20; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_conv.py -m gpu -x -q -p no:cacheprovider -k "WR_ or weight_resident or pipelined" > $O/tests.log 2>&1; tail -5 $O/tests.log
for dbg in 0 1 2 32 35; do
for cfg in "16 9 60 256 7 1 3,3,3,3 fwd" "16 60 3 256 7 1 3,3,3,3 fwd" "16 60 3 256 7 1 3,3,3,3 bwd" "32 15 64 256 4 2 1,1,1,1 fwd"; do
  set -- $cfg
  echo -n "DBG=$dbg C$2 K$3 H$4 R$5 s$6 $8: " >> $O/ab.log
  HIFIC_DBG=$dbg MPROF=1 MN=$1 MC=$2 MK=$3 MH=$4 MR=$5 MS=$6 MPAD=$7 timeout 120 python tools/micro_conv.py $8 20 2>&1 | grep -E "^(fwd|bwd)" | sed 's/^[^[]*\[/[/' >> $O/ab.log
done; done
cat $O/ab.log
