R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_run45; mkdir -p $O; cd $R
for lib in qb1 default; do
  L=""; [ $lib = qb1 ] && L=$R/high-fidelity-generative-compression_amd/libhific_hip_qb1.so
  for cfg in "16 3 60" "16 60 3"; do set -- $cfg
    echo -n "$lib wgrad C$2 K$3: " >> $O/ab.log
    HIFIC_LIB_PATH=$L MPROF=1 MN=$1 MC=$2 MK=$3 MH=256 MR=7 MS=1 timeout 120 python tools/micro_conv.py wgrad 20 2>&1 | grep -E "^bwd_weight" | sed 's/^[^[]*\[/[/' >> $O/ab.log
  done
done
cat $O/ab.log
