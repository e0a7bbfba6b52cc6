# Round-end measurement run (GPU box): tests, bench (+CPU baseline), rocprofv3 kernel stats, PMC traffic passes.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final; mkdir -p $O
cd $R
timeout 400 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; tail -2 $O/tests.log
timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; cut -c1-300 $O/bench.json
timeout 120 python bench.py --config gan --steps 8 --warmup 3 --no-cpu-baseline > $O/bench_gan.json 2>> $O/bench.err; cut -c1-200 $O/bench_gan.json
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/ks -o ks -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > /tmp/ks.log 2>&1
db=$(find /tmp/ks -name "*.db" | head -1); [ -n "$db" ] && python $R/tools/rocpd_stats.py $db > $O/kernel_stats.md 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pf -o pf -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > /tmp/pf.log 2>&1 || echo "fetch pass failed/timeout"
timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/pw -o pw -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > /tmp/pw.log 2>&1 || echo "write pass failed/timeout"
dbf=$(find /tmp/pf -name "*.db" | head -1); dbw=$(find /tmp/pw -name "*.db" | head -1)
[ -n "$dbf" ] && [ -n "$dbw" ] && python $R/tools/collect_traffic.py $dbf $dbw $O/pmc_traffic.json
[ -n "$dbf" ] && python $R/tools/rocpd_pmc.py $dbf > $O/pmc_fetch.txt 2>&1
[ -n "$dbw" ] && python $R/tools/rocpd_pmc.py $dbw > $O/pmc_write.txt 2>&1
echo done
