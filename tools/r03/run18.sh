# where does the residual-block kernel's time go: timing ablations (SP9_ABL builds, wrong results by construction)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_run18; mkdir -p $O; cd $R
for w in 0 1; do
  HIFIC_SP9_W4=$w timeout 120 python tools/micro_sp9.py 40 2>/dev/null | sed "s/^/abl=0 /"
  for a in 1 2 3 4 7 15; do
    HIFIC_SP9_W4=$w HIFIC_LIB_PATH=$R/gpurun_ab/libhific_abl$a.so timeout 120 python tools/micro_sp9.py 40 2>/dev/null | sed "s/^/abl=$a /"
  done
done 2>&1 | tee $O/abl.log
