R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_run4; mkdir -p $O; cd $R
timeout 900 python tools/r06/sweep_exact.py > $O/sweep.log 2>&1; grep sweep $O/sweep.log; tail -3 $O/sweep.log
