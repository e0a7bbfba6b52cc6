R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_run46; mkdir -p $O; cd $R
for tgt in 512 768 1024 1536 2048; do
  for cfg in "16 3 60" "16 60 3"; do set -- $cfg
    echo -n "TARGET=$tgt wgrad C$2 K$3: " >> $O/ab.log
    HIFIC_IM2COL_TARGET=$tgt MPROF=1 MN=$1 MC=$2 MK=$3 MH=256 MR=7 MS=1 timeout 120 python tools/micro_conv.py wgrad 20 2>&1 | grep -E "^bwd_weight" | sed 's/(.*pack)//' >> $O/ab.log
  done
done
cat $O/ab.log
