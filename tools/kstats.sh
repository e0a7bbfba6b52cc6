cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/ks -o ks -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline > /tmp/ks.log 2>&1
tail -2 /tmp/ks.log | cut -c1-200
db=$(find /tmp/ks -name "*.db" | head -1)
python $R/tools/rocpd_stats.py $db > $R/gpurun_out/kstats_now.md 2>&1
