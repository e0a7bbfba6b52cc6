# timeline of the eager multi-stream cycle: GPU busy vs idle gaps, per queue; plus full kernel stats (all kernels)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_run9; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
HIFIC_BENCH_GRAPH=0 timeout 300 rocprofv3 --kernel-trace -d /tmp/tl -o tl -- python $R/bench.py --steps 6 --warmup 3 --no-extras > /tmp/tl.log 2>&1
db=$(find /tmp/tl -name "*.db" | head -1)
python $R/tools/timeline.py $db 0.6 > $O/timeline.txt 2>&1
python $R/tools/rocpd_stats.py $db > $O/stats_multistream.md 2>&1
tail -2 /tmp/tl.log | cut -c1-200
head -45 $O/timeline.txt
