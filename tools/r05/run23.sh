R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_run23; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
export HIFIC_SIDE_WGRAD=0 HIFIC_BRANCH_STREAMS=0 HIFIC_BENCH_GRAPH=0
timeout 250 rocprofv3 --kernel-trace --stats -d /tmp/ksg -o ks -- python $R/bench.py --steps 5 --warmup 2 --no-extras > /tmp/ksg.log 2>&1
dbg=$(find /tmp/ksg -name "*.db" | head -1)
[ -n "$dbg" ] && python $R/tools/rocpd_stats.py $dbg > $O/kernel_stats_gan.md 2>&1
tail -3 /tmp/ksg.log | cut -c1-300
head -5 $O/kernel_stats_gan.md | cut -c1-200
