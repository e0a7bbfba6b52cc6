# Round 2, GPU call 1: all GPU tests, the new bench line (with traffic child runs + CPU baseline), 1-rank RCCL run,
# rocprofv3 kernel stats of the headline.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02_run1; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q -s -p no:cacheprovider > $O/tests.log 2>&1; tail -5 $O/tests.log
grep -E "^\[|flips|FAILED|ERROR" $O/tests.log | head -40
timeout 420 python bench.py --steps 8 --warmup 3 > $O/bench.json 2> $O/bench.err; cut -c1-600 $O/bench.json; tail -3 $O/bench.err
HIFIC_FORCE_DIST=1 timeout 200 python bench.py --steps 3 --warmup 2 --no-extras > $O/rccl_1rank.log 2>&1; tail -2 $O/rccl_1rank.log | cut -c1-300
cd /tmp && export TMPDIR=/tmp
timeout 250 rocprofv3 --kernel-trace --stats -d /tmp/ks -o ks -- python $R/bench.py --steps 5 --warmup 2 --no-extras > /tmp/ks.log 2>&1
db=$(find /tmp/ks -name "*.db" | head -1); [ -n "$db" ] && python $R/tools/rocpd_stats.py $db > $O/kernel_stats_gan.md 2>&1
timeout 250 rocprofv3 --kernel-trace --stats -d /tmp/ks2 -o ks -- python $R/bench.py --config compression --steps 5 --warmup 2 --no-extras > /tmp/ks2.log 2>&1
db=$(find /tmp/ks2 -name "*.db" | head -1); [ -n "$db" ] && python $R/tools/rocpd_stats.py $db > $O/kernel_stats_compression.md 2>&1
echo done
