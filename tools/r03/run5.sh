R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_run5; mkdir -p $O; cd $R
timeout 1200 python tools/graph_probe.py > $O/probe.log 2>&1; cat $O/probe.log
