# the bench contract: exactly one JSON line on stdout, also with RCCL in the process
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_run30; mkdir -p $O; cd $R
HIFIC_FORCE_DIST=1 timeout 300 python bench.py --steps 3 --warmup 2 --no-extras > $O/dist.out 2> $O/dist.err
echo "dist: $(wc -l < $O/dist.out) stdout line(s), banner lines on stderr: $(grep -c 'RCCL version' $O/dist.err)"; head -c 120 $O/dist.out; echo
timeout 300 python bench.py --steps 3 --warmup 2 --no-extras > $O/plain.out 2> $O/plain.err
echo "plain: $(wc -l < $O/plain.out) stdout line(s)"; head -c 120 $O/plain.out; echo
timeout 600 python bench.py --gpus 1 --steps 4 --warmup 2 > $O/full.out 2> $O/full.err
echo "full: $(wc -l < $O/full.out) stdout line(s)"; python -c "import json;d=json.loads(open('$O/full.out').read());print(d['value'], sorted(d.keys()))"
