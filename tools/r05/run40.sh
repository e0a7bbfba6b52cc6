R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_run40; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests/test_gpu_conv.py -m gpu -x -q -p no:cacheprovider -k "PM_ or merged_phase" > $O/tests.log 2>&1; tail -3 $O/tests.log
for dbg in 0 1 32; do
  echo -n "DBG=$dbg convT C120 K60 H128: " >> $O/ab.log
  HIFIC_DBG=$dbg MPROF=1 MN=16 MC=120 MK=60 MH=128 timeout 120 python tools/micro_conv.py convt 20 2>&1 | grep -E "^convt" | sed 's/^[^[]*\[/[/' >> $O/ab.log
  echo -n "DBG=$dbg convT C240 K120 H64: " >> $O/ab.log
  HIFIC_DBG=$dbg MPROF=1 MN=16 MC=240 MK=120 MH=64 timeout 120 python tools/micro_conv.py convt 20 2>&1 | grep -E "^convt" | sed 's/^[^[]*\[/[/' >> $O/ab.log
  echo -n "DBG=$dbg dgrad C128 K256 H64 R4: " >> $O/ab.log
  HIFIC_DBG=$dbg MPROF=1 MN=32 MC=128 MK=256 MH=64 MR=4 MS=2 MPAD=1,1,1,1 timeout 120 python tools/micro_conv.py bwd 20 2>&1 | grep -E "^bwd" | sed 's/^[^[]*\[/[/' >> $O/ab.log
  echo -n "DBG=$dbg dgrad C60 K120 H256 R3: " >> $O/ab.log
  HIFIC_DBG=$dbg MPROF=1 MN=16 MC=60 MK=120 MH=256 MR=3 MS=2 MPAD=1,0,0,1 timeout 120 python tools/micro_conv.py bwd 20 2>&1 | grep -E "^bwd" | sed 's/^[^[]*\[/[/' >> $O/ab.log
done
cat $O/ab.log
