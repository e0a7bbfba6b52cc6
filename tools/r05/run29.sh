R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_run29; mkdir -p $O; cd $R
timeout 300 python tools/trace_aten.py > $O/aten.log 2>&1
grep -n "=== by stack" -A60 $O/aten.log | cut -c1-260
