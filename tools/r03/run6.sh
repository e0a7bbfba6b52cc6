R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_run6; mkdir -p $O; cd $R
timeout 1200 python tools/graph_probe.py syn_fork_alloc syn_fork_twice syn_record_stream syn_event_later syn_fork_hific model_fwd_mask1 model_fwd_mask2 model_fwd_mask4 model_fwd_mask3 gturn_mask1 gturn_mask2 gturn_mask4 > $O/probe.log 2>&1; cat $O/probe.log
