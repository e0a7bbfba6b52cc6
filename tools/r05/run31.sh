R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_run31; mkdir -p $O; cd $R
for tgt in 800 512 384 256; do
for cfg in "16 2880 220 16" "16 960 220 16" "16 320 220 16" "16 660 320 16"; do set -- $cfg
  echo -n "TARGET=$tgt C$2 K$3 H$4: " >> $O/ks.log
  HIFIC_KSPLIT_TARGET=$tgt MPROF=1 MN=$1 MC=$2 MK=$3 MH=$4 MR=3 MS=1 timeout 120 python tools/micro_conv.py fwd 30 2>&1 | grep -E "^fwd" | sed 's/^fwd: //' >> $O/ks.log
done; done
cat $O/ks.log
