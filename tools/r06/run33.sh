# does the rehearsal catch the bug it was written for?  bench.py with the rank-0-alone pass issuing the q_bpp collective again
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_run33; mkdir -p $O; cd $R
sed 's/scalars_were = parallel.set_scalar_collectives(False)/scalars_were = parallel.set_scalar_collectives(True)/' bench.py > bench_old.py
grep -c "set_scalar_collectives(True)" bench_old.py
HIFIC_BENCH_REHEARSAL=1 timeout 240 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29631 bench_old.py --gpus 2 --steps 2 --warmup 1 > $O/old.out 2> $O/old.err
echo "old code: rc=$? (124 = killed by timeout), stdout lines: $(wc -l < $O/old.out)"
tail -4 $O/old.err | cut -c1-300
rm -f bench_old.py
