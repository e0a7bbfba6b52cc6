R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_run27; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/kd1 -o kd -- python $R/tools/r05/micro_d1.py > /tmp/kd1.log 2>&1
db=$(find /tmp/kd1 -name "*.db" | head -1)
python $R/tools/rocpd_stats.py $db 2>&1 | head -8 | cut -c1-160 | tee $O/stats.md
