R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_run51; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_conv.py -m gpu -x -q -p no:cacheprovider -k "WC3_" > $O/tests.log 2>&1; tail -2 $O/tests.log
for dbg in 0 1 2 3; do
  echo -n "DBG=$dbg wgrad C3 K60: " >> $O/ab.log
  HIFIC_DBG=$dbg MPROF=1 MN=16 MC=3 MK=60 MH=256 MR=7 MS=1 timeout 120 python tools/micro_conv.py wgrad 20 2>&1 | grep -E "^bwd_weight" | sed 's/(.*pack)//' >> $O/ab.log
done
cat $O/ab.log
