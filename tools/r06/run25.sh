R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_run25; mkdir -p $O; cd $R
timeout 600 python tools/host_floor_probe.py profile > $O/profile.log 2>&1
timeout 300 python tools/host_floor_probe.py eager > $O/eager.log 2>&1
tail -3 $O/eager.log
