R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_run32; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
export HIFIC_BENCH_GRAPH=0
timeout 250 rocprofv3 --kernel-trace -d /tmp/tl -o tl -- python $R/bench.py --steps 6 --warmup 2 --no-extras > /tmp/tl.log 2>&1
db=$(find /tmp/tl -name "*.db" | head -1)
python $R/tools/timeline.py $db 0.6 > $O/timeline.txt 2>&1
head -60 $O/timeline.txt | cut -c1-220
