R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_run33; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
export HIFIC_BENCH_GRAPH=0
timeout 250 rocprofv3 --kernel-trace -d /tmp/tl -o tl -- python $R/bench.py --steps 8 --warmup 3 --no-extras > /tmp/tl.log 2>&1
db=$(find /tmp/tl -name "*.db" | head -1)
python $R/tools/r05/gaps.py $db 150 2400 > $O/gaps.txt 2>&1
head -120 $O/gaps.txt | cut -c1-170
