R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02_run20; mkdir -p $O
cd $R
export MN=16 MC=60 MK=120 MH=256 MR=3 MS=2
for ws in 0 1; do
 for dbg in 0 64 32 96 2 4 6 102; do
  echo "fwd WSTAGE=$ws DBG=$dbg: $(HIFIC_WSTAGE=$ws HIFIC_DBG=$dbg timeout 120 python tools/micro_conv.py fwd 30 2>&1 | tail -1)"
 done
 for dbg in 0 1 2 3 4 7; do
  echo "wgrad WSTAGE=$ws DBG=$dbg: $(HIFIC_WSTAGE=$ws HIFIC_DBG=$dbg timeout 120 python tools/micro_conv.py wgrad 30 2>&1 | tail -1)"
 done
done 2>&1 | tee $O/ablate.txt
