R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_run14; mkdir -p $O; cd $R
P=$R/high-fidelity-generative-compression_amd
for cfg in "16 480 960 32" "16 60 120 256"; do
  set -- $cfg
  for abl in 0 63 47 55; do
    lib=$P/libhific_hip_abl$abl.so; [ $abl = 0 ] && lib=$P/libhific_hip.so
    echo -n "C$2 K$3 H$4 abl=$abl: " >> $O/abl.log
    HIFIC_LIB_PATH=$lib MPROF=1 MN=$1 MC=$2 MK=$3 MH=$4 MR=3 MS=2 timeout 120 python tools/micro_conv.py fwd 20 2>&1 | grep "^fwd" | sed 's/.*\[//' >> $O/abl.log
  done
done
cat $O/abl.log
