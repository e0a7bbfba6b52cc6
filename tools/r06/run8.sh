# host enqueue time of the eager cycle, same box: round-6 host-path changes on / off
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_run8; mkdir -p $O; cd $R
for rep in 1 2; do
  timeout 200 python tools/host_floor_probe.py eager 2>&1 | grep probe | sed "s/^/[all on] /"
  HIFIC_CONV_NORM_FUSED=0 timeout 200 python tools/host_floor_probe.py eager 2>&1 | grep probe | sed "s/^/[conv_norm off] /"
  HIFIC_FUSED_LOSS=0 timeout 200 python tools/host_floor_probe.py eager 2>&1 | grep probe | sed "s/^/[fused loss off] /"
  HIFIC_CONV_NORM_FUSED=0 HIFIC_FUSED_LOSS=0 HIFIC_TICKETS=0 timeout 200 python tools/host_floor_probe.py eager 2>&1 | grep probe | sed "s/^/[all off] /"
done > $O/host.log 2>&1
cat $O/host.log
timeout 300 python tools/host_floor_probe.py profile > $O/hostprof.txt 2>&1; head -45 $O/hostprof.txt
