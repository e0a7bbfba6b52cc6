R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_run35; mkdir -p $O; cd $R
for g in 256 64 8; do
for cfg in "16 320 320 16" "16 320 320 8" "16 640 320 16"; do set -- $cfg
  echo -n "BIGTILE_MIN_GRID=$g C$2 K$3 H$4: " >> $O/bt.log
  HIFIC_PROF_DUMP=1 HIFIC_GC_BIGTILE_MIN_GRID=$g MPROF=1 MN=$1 MC=$2 MK=$3 MH=$4 MR=5 MS=2 MPAD=2,2,2,2 timeout 120 python tools/micro_conv.py all 30 2> $O/dump.txt | grep -E "^(fwd|bwd)" | sed 's/(.*incl. pack)//' | tr '\n' ' ' >> $O/bt.log
  grep HIFIC_PROF $O/dump.txt | awk '{k=$0; sub(/^HIFIC_PROF [^ ]+ [0-9.]+ [0-9.e+]+ /,"",k); print k}' | sort -u | tr '\n' ';' | cut -c1-400 >> $O/bt.log
  echo >> $O/bt.log
done; done
cat $O/bt.log
