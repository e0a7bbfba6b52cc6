R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02_run7; mkdir -p $O
cd $R
for shape in "60 120 256" "240 480 64"; do
  set -- $shape
  for big in 1 0; do
  for dbg in 0 2 34 66 98 4 16; do
    echo "== C=$1 K=$2 H=$3 s2 NO_BIGSTAGE=$big DBG=$dbg"
    HIFIC_NO_BIGSTAGE=$big MC=$1 MK=$2 MH=$3 MS=2 HIFIC_DBG=$dbg timeout 120 python tools/micro_conv.py fwd 20 2>&1 | grep -v "Warn\|amdgpu.ids"
  done
  done
done > $O/micro.txt 2>&1
cat $O/micro.txt
