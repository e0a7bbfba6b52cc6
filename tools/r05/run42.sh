R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_run42; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
export HIFIC_BENCH_GRAPH=0
timeout 250 rocprofv3 --kernel-trace -d /tmp/tl -o tl -- python $R/bench.py --steps 8 --warmup 3 --no-extras > /tmp/tl.log 2>&1
db=$(find /tmp/tl -name "*.db" | head -1)
python $R/tools/r05/tail.py $db 1500 > $O/tail.txt 2>&1
head -150 $O/tail.txt | cut -c1-150
