R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_run14; mkdir -p $O; cd $R
timeout 300 python tools/trace_aten.py > $O/aten.log 2>&1; grep -v "^-" $O/aten.log | cut -c1-230 | head -130
