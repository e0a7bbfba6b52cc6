# round 6, call 3: kernel trace of the cycle, plain bf16 vs exact training (fused chain), single stream
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_run3; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export HIFIC_SIDE_WGRAD=0 HIFIC_BRANCH_STREAMS=0 HIFIC_BENCH_GRAPH=0
for m in 1 0; do
  HIFIC_EXACT_TRAIN=$m timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/ks$m -o ks -- python $R/bench.py --steps 5 --warmup 2 --no-extras > /tmp/ks$m.log 2>&1
  db=$(find /tmp/ks$m -name "*.db" | head -1)
  [ -n "$db" ] && python $R/tools/rocpd_stats.py $db > $O/kstats_exact$m.md 2>&1
  tail -1 /tmp/ks$m.log | cut -c1-200
done
echo done
