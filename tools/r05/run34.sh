R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_run34; mkdir -p $O; cd $R
timeout 300 python tools/r05/drain_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/drain.txt
