R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_run53; mkdir -p $O; cd $R
timeout 200 python tools/host_floor_probe.py eager 2>&1 | grep probe | tee -a $O/g.log
timeout 300 python tools/host_floor_probe.py graph 2>&1 | grep probe | tee -a $O/g.log
timeout 300 python tools/host_floor_probe.py graph 1stream 2>&1 | grep probe | tee -a $O/g.log
