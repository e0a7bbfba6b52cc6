R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_run25; mkdir -p $O; cd $R
timeout 100 python tools/r05/micro_d1.py > $O/d1.log 2>&1; cat $O/d1.log
for cfg in "32 15 64" "16 3 64"; do set -- $cfg
echo -n "dgrad N$1 C$2 K$3: " >> $O/d1.log
MPROF=1 MN=$1 MC=$2 MK=$3 MH=256 MR=4 MS=2 MPAD=1,1,1,1 timeout 120 python tools/micro_conv.py bwd 20 2>&1 | grep -E "^bwd" >> $O/d1.log
done; cat $O/d1.log
