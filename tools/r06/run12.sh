# re-test of two round-2 stream knobs at the round-6 balance + FETCH_SIZE calibration
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_run12; mkdir -p $O; cd $R
for rep in 1 2; do
  for v in "base" "HIFIC_PACK_STREAM=1" "HIFIC_OPT_STREAM=1" "HIFIC_SIDE_STREAMS=2"; do
    if [ "$v" = base ]; then e=""; else e="$v"; fi
    env $e HIFIC_BENCH_GRAPH=0 timeout 200 python bench.py --steps 12 --warmup 4 --no-extras > $O/b.json 2>$O/b.err
    python -c "import json;d=json.loads([l for l in open('$O/b.json') if l.startswith('{')][0]);print('$v rep $rep:', d['ms_per_step'], 'ms', d['value'], 'img/s')"
  done
done
bash tools/ubench/fetchcal.sh
