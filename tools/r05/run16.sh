R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_run16; mkdir -p $O; cd $R
P=$R/high-fidelity-generative-compression_amd
for rep in 1 2; do
for v in base spr; do
  lib=$P/libhific_hip_spr.so; [ $v = base ] && lib=$P/libhific_hip.so
  for cfg in "16 480 960 32" "16 60 120 256" "16 240 480 64"; do
    set -- $cfg
    echo -n "$v C$2 K$3 H$4: " >> $O/spr.log
    HIFIC_LIB_PATH=$lib MPROF=1 MN=$1 MC=$2 MK=$3 MH=$4 MR=3 MS=2 timeout 120 python tools/micro_conv.py fwd 20 2>&1 | grep "^fwd" | sed 's/.*\[//' >> $O/spr.log
  done
done
done
cat $O/spr.log
