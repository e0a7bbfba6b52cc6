R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_run14; mkdir -p $O; cd $R
python -c "import torch; print('priority range', torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream,'priority_range') else None)"
for rep in 1 2 3; do
  for v in "base" "HIFIC_SIDE_PRIO=1" "HIFIC_SIDE_PRIO=-1" "HIFIC_SIDE_PRIO=1 HIFIC_PACK_STREAM=1"; do
    if [ "$v" = base ]; then e=""; else e="$v"; fi
    env $e HIFIC_BENCH_GRAPH=0 timeout 200 python bench.py --steps 12 --warmup 4 --no-extras > $O/b.json 2>$O/b.err
    python -c "import json;d=json.loads([l for l in open('$O/b.json') if l.startswith('{')][0]);print('$v rep $rep:', d['ms_per_step'], 'ms', d['value'], 'img/s')" 2>&1 | tail -1
  done
done
