R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_run29; mkdir -p $O; cd $R
timeout 600 python tools/soak.py 600 > $O/soak.log 2>&1; tail -6 $O/soak.log | cut -c1-400
