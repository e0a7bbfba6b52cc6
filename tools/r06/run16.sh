R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_run16; mkdir -p $O; cd $R
for rep in 1 2 3 4; do
  for v in "base" "HIFIC_SIDE_STREAMS=2" "HIFIC_SIDE_STREAMS=3"; do
    if [ "$v" = base ]; then e=""; else e="$v"; fi
    env $e HIFIC_BENCH_GRAPH=0 timeout 200 python bench.py --steps 16 --warmup 4 --no-extras > $O/b.json 2>$O/b.err
    python -c "import json;d=json.loads([l for l in open('$O/b.json') if l.startswith('{')][0]);print('$v rep $rep:', d['ms_per_step'], 'ms', d['value'], 'img/s')" 2>&1 | tail -1
  done
done
