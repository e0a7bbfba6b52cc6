R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_run39; mkdir -p $O; cd $R
for dbg in 0 1 2 32 35; do
  echo -n "DBG=$dbg convT C120 K60 H128: " >> $O/ab.log
  HIFIC_DBG=$dbg MPROF=1 MN=16 MC=120 MK=60 MH=128 timeout 120 python tools/micro_conv.py convt 20 2>&1 | grep -E "^convt" | sed 's/^[^[]*\[/[/' >> $O/ab.log
  echo -n "DBG=$dbg dgrad C128 K256 H64 R4: " >> $O/ab.log
  HIFIC_DBG=$dbg MPROF=1 MN=32 MC=128 MK=256 MH=64 MR=4 MS=2 MPAD=1,1,1,1 timeout 120 python tools/micro_conv.py bwd 20 2>&1 | grep -E "^bwd" | sed 's/^[^[]*\[/[/' >> $O/ab.log
done
cat $O/ab.log
