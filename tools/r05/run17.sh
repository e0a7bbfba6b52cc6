R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_run17; mkdir -p $O; cd $R
P=$R/high-fidelity-generative-compression_amd
for rep in 1 2; do for v in 0 1 2; do
  lib=$P/libhific_hip_ad$v.so; [ $v = 0 ] && lib=$P/libhific_hip.so
  echo -n "ADAM_V=$v: " >> $O/adam.log; HIFIC_LIB_PATH=$lib timeout 120 python tools/r05/micro_adam.py 2>&1 | grep "^adam" >> $O/adam.log
done; done
cat $O/adam.log
