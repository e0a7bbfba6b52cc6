R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02_run33; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 250 rocprofv3 --kernel-trace -d /tmp/tl -o tl -- python $R/bench.py --steps 6 --warmup 3 --no-extras > /tmp/tl.log 2>&1
db=$(find /tmp/tl -name "*.db" | head -1)
python $R/tools/timeline.py $db 0.5 > $O/timeline_gan.txt 2>&1; head -50 $O/timeline_gan.txt
