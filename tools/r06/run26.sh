cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_run26; mkdir -p $O
timeout 250 rocprofv3 --kernel-trace -d /tmp/tl -o tl -- python $R/tools/host_floor_probe.py eager > /tmp/tl.log 2>&1
tail -1 /tmp/tl.log
db=$(find /tmp/tl -name "*.db" | head -1)
python $R/tools/timeline.py $db 0.75 > $O/timeline_multi.txt 2>&1
tail -42 $O/timeline_multi.txt | cut -c1-200
