# Round-6 measurement run (GPU box): all GPU tests, smoke, the bench line (all legs, live traffic, CPU baseline), 1-rank RCCL
# runs (f32 and bf16 gradient payload), rocprofv3 kernel trace of the headline config, per-kernel HBM / MFMA-busy table
# (separate --pmc passes; kernel-trace only, single-stream eager launching so that every dispatch is timed alone).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${OUTTAG:-r06_final}; mkdir -p $O
cd $R
git rev-parse HEAD > $O/head.txt 2>/dev/null
timeout 1500 python -m pytest tests -m gpu -q -s -p no:cacheprovider > $O/gpu_tests.log 2>&1; tail -3 $O/gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
HIFIC_BENCH_DIAG=1 timeout 1200 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; cut -c1-300 $O/bench.json
HIFIC_FORCE_DIST=1 timeout 200 python bench.py --steps 3 --warmup 2 --no-extras > $O/rccl_1rank.log 2>&1; tail -1 $O/rccl_1rank.log | cut -c1-200
HIFIC_FORCE_DIST=1 HIFIC_GRAD_PAYLOAD=bf16 timeout 200 python bench.py --steps 3 --warmup 2 --no-extras > $O/rccl_1rank_bf16.log 2>&1; tail -1 $O/rccl_1rank_bf16.log | cut -c1-200
cd /tmp && export TMPDIR=/tmp
export HIFIC_SIDE_WGRAD=0 HIFIC_BRANCH_STREAMS=0 HIFIC_BENCH_GRAPH=0
timeout 250 rocprofv3 --kernel-trace --stats -d /tmp/ksg -o ks -- python $R/bench.py --steps 5 --warmup 2 --no-extras > /tmp/ksg.log 2>&1
dbg=$(find /tmp/ksg -name "*.db" | head -1)
[ -n "$dbg" ] && python $R/tools/rocpd_stats.py $dbg > $O/kernel_stats_gan.md 2>&1
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set -d /tmp/pm$i -o pm -- python $R/bench.py --steps 5 --warmup 2 --no-extras > /tmp/pm$i.log 2>&1 || echo "pmc pass $i failed/timeout"
done
d1=$(find /tmp/pm1 -name "*.db" | head -1); d2=$(find /tmp/pm2 -name "*.db" | head -1); d3=$(find /tmp/pm3 -name "*.db" | head -1)
[ -n "$dbg" ] && [ -n "$d1" ] && [ -n "$d2" ] && [ -n "$d3" ] && python $R/tools/kernel_table.py $dbg $d1 $d2 $d3 2 > $O/kernel_table_gan.md 2>&1
head -20 $O/kernel_table_gan.md | cut -c1-200
echo done
