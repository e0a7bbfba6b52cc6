R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_run41; mkdir -p $O; cd $R
for bc in 64 32; do
for cfg in "16 120 60 128" "16 240 120 64"; do set -- $cfg
  echo -n "MP_BC=$bc convT C$2 K$3 H$4: " >> $O/ab.log
  HIFIC_MP_BC=$bc MPROF=1 MN=$1 MC=$2 MK=$3 MH=$4 timeout 120 python tools/micro_conv.py convt 20 2>&1 | grep -E "^convt" | sed 's/^[^[]*\[/[/' >> $O/ab.log
done
for cfg in "32 64 128 128 4 1,1,1,1" "32 128 256 64 4 1,1,1,1" "32 256 512 32 4 1,1,1,1" "16 60 120 256 3 1,0,0,1" "16 120 240 128 3 1,0,0,1"; do set -- $cfg
  echo -n "MP_BC=$bc dgrad C$2 K$3 H$4 R$5: " >> $O/ab.log
  HIFIC_MP_BC=$bc MPROF=1 MN=$1 MC=$2 MK=$3 MH=$4 MR=$5 MS=2 MPAD=$6 timeout 120 python tools/micro_conv.py bwd 20 2>&1 | grep -E "^bwd" | sed 's/^[^[]*\[/[/' >> $O/ab.log
done; done
cat $O/ab.log
HIFIC_MP_BC=32 timeout 900 python -m pytest tests/test_gpu_conv.py -m gpu -x -q -p no:cacheprovider -k "PM_ or U4 or U1 or S2T or E2_ or D4_" > $O/tests.log 2>&1; tail -3 $O/tests.log
