R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02_run4; mkdir -p $O
cd $R
for shape in "60 120 256" "120 240 128" "240 480 64" "480 960 32"; do
  set -- $shape
  for dbg in 0 1 2 3; do
    echo "== C=$1 K=$2 H=$3 s2 DBG=$dbg"
    MC=$1 MK=$2 MH=$3 MS=2 HIFIC_DBG=$dbg timeout 120 python tools/micro_conv.py all 20 2>&1 | grep -v Warn
  done
done > $O/micro.txt 2>&1
cat $O/micro.txt
