R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02_run14; mkdir -p $O
cd $R
for tps in 1 0; do
for dbg in 0 98 102 114 118; do
  echo "== G9 fwd K3 C60 7x7 NO_TPS=$tps DBG=$dbg"
  HIFIC_NO_TPS=$tps MC=60 MK=3 MH=256 MR=7 MS=1 HIFIC_DBG=$dbg timeout 120 python tools/micro_conv.py fwd 10 2>&1 | grep -v "Warn\|amdgpu.ids"
done
done > $O/micro.txt 2>&1
cat $O/micro.txt
