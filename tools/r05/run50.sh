R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_run50; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
MN=16 MC=3 MK=60 MH=256 MR=7 MS=1 timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/kc3 -o kc -- python $R/tools/micro_conv.py wgrad 20 > /tmp/kc3.log 2>&1
db=$(find /tmp/kc3 -name "*.db" | head -1)
python $R/tools/rocpd_stats.py $db 2>&1 | head -6 | cut -c1-150 | tee $O/stats.md
