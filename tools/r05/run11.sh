R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_run11; mkdir -p $O; cd $R
P=$R/high-fidelity-generative-compression_amd
for rep in 1 2; do
for v in 0 1 2 3; do
  lib=$P/libhific_hip_spx$v.so; [ $v = 0 ] && lib=$P/libhific_hip.so
  echo "== variant $v" >> $O/sp9.log
  HIFIC_LIB_PATH=$lib timeout 120 python tools/micro_sp9.py 40 2>&1 | grep "gconv_sp9" >> $O/sp9.log
done
done
cat $O/sp9.log
