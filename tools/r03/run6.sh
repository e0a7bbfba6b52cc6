R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_run6; mkdir -p $O; cd $R
timeout 1200 python tools/graph_probe.py model_fwd_mask5 model_fwd_mask6 model_fwd_mask7 gturn_mask7 gturn_all cycle_all > $O/probe2.log 2>&1; cat $O/probe2.log
