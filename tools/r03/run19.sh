# prefetch distances of the A-from-global loop (SP9_PD / SP9_PFIRST / SP9_W4_PFIRST builds), kernel-only timing
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_run19; mkdir -p $O; cd $R
for rep in 1 2; do
for w in 0 1; do
  HIFIC_SP9_W4=$w timeout 120 python tools/micro_sp9.py 40 2>/dev/null | sed "s/^/base /"
  for a in A B E F; do
    HIFIC_SP9_W4=$w HIFIC_LIB_PATH=$R/gpurun_ab/libhific_$a.so timeout 120 python tools/micro_sp9.py 40 2>/dev/null | sed "s/^/$a /"
  done
done
done 2>&1 | sed 's/HIFIC_LIB_PATH=[^ ]* //' | tee $O/pd.log
