R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02_run11; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 250 rocprofv3 --kernel-trace --stats -d /tmp/ks -o ks -- python $R/bench.py --config compression --steps 5 --warmup 2 --no-extras > /tmp/ks.log 2>&1
db=$(find /tmp/ks -name "*.db" | head -1); [ -n "$db" ] && python $R/tools/rocpd_stats.py $db > $O/kernel_stats_compression.md 2>&1
head -30 $O/kernel_stats_compression.md | cut -c1-140
