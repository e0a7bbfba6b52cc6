# what do the dispatch gaps of eager launching cost?  same kernels, one stream: eager vs graph replay; and the multi-stream graph
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_run28; mkdir -p $O; cd $R
for rep in 1 2; do
timeout 300 python tools/host_floor_probe.py eager 1stream 2>&1 | grep probe | tee -a $O/ab.log
timeout 300 python tools/host_floor_probe.py graph 1stream 2>&1 | grep "end-to-end" | tee -a $O/ab.log
timeout 300 python tools/host_floor_probe.py eager 2>&1 | grep probe | tee -a $O/ab.log
timeout 300 python tools/host_floor_probe.py graph 2>&1 | grep "end-to-end" | tee -a $O/ab.log
done
