R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_run55; mkdir -p $O; cd $R
for bm in 0 64 128; do
for cfg in "16 320 320 16" "16 320 320 8" "16 640 320 16"; do set -- $cfg
  echo -n "HIFIC_BM=$bm C$2 K$3 H$4: " >> $O/bm.log
  HIFIC_BM=$bm MPROF=1 MN=$1 MC=$2 MK=$3 MH=$4 MR=5 MS=2 MPAD=2,2,2,2 timeout 120 python tools/micro_conv.py all 30 2>/dev/null | grep -E "^(fwd|bwd_data)" | sed 's/(.*incl. pack)//' | tr '\n' ' ' >> $O/bm.log
  echo >> $O/bm.log
done; done
cat $O/bm.log
