R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_run49; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_conv.py -m gpu -x -q -s -p no:cacheprovider -k "WC3_ or WR_7x7_c3 or strided_weight" > $O/tests.log 2>&1; tail -4 $O/tests.log; grep "WC3" $O/tests.log | head
for c3 in 0 1; do
  echo -n "WGRAD_C3=$c3 wgrad C3 K60: " >> $O/ab.log
  HIFIC_WGRAD_C3=$c3 MPROF=1 MN=16 MC=3 MK=60 MH=256 MR=7 MS=1 timeout 120 python tools/micro_conv.py wgrad 20 2>&1 | grep -E "^bwd_weight" | sed 's/(.*pack)//' >> $O/ab.log
done
cat $O/ab.log
