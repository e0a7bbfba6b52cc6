R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_run10; mkdir -p $O; cd $R
run() { echo -n "$1 | $2: " >> $O/sweep.log; env $1 MPROF=1 $2 timeout 120 python tools/micro_conv.py fwd 20 2>&1 | grep "^fwd" | sed 's/.*\[//' >> $O/sweep.log; }
for shape in "MN=16 MC=60 MK=120 MH=256 MR=3 MS=2" "MN=16 MC=120 MK=240 MH=128 MR=3 MS=2" "MN=16 MC=240 MK=480 MH=64 MR=3 MS=2" "MN=16 MC=480 MK=960 MH=32 MR=3 MS=2" "MN=32 MC=64 MK=128 MH=128 MR=4 MS=2 MPAD=1,1,1,1" "MN=32 MC=256 MK=512 MH=32 MR=4 MS=2 MPAD=1,1,1,1" "MN=32 MC=15 MK=64 MH=256 MR=4 MS=2 MPAD=1,1,1,1"; do
  for knob in "X=0" "HIFIC_PL_TW=16" "HIFIC_PL_WSHARE_KB=0" "HIFIC_PL_WSHARE_KB=100000000" "HIFIC_PL_TG=2"; do
    run "$knob" "$shape"
  done
done
cat $O/sweep.log
