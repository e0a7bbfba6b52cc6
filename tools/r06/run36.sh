# `python bench.py --gpus 2` from a plain shell (bench.py spawns its own ranks), rehearsed on one GPU
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_run36; mkdir -p $O; cd $R
HIFIC_BENCH_REHEARSAL=1 timeout 600 python bench.py --gpus 2 --steps 2 --warmup 1 > $O/respawn.out 2> $O/respawn.err
echo "respawn rc=$? stdout lines: $(wc -l < $O/respawn.out)"; head -c 200 $O/respawn.out; echo
