# runtime dispatch knobs (not library code): kernarg placement, hardware queue count
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_run27; mkdir -p $O; cd $R
for rep in 1 2 3; do
  for v in "base" "HIP_FORCE_DEV_KERNARG=1" "HIP_FORCE_DEV_KERNARG=0" "GPU_MAX_HW_QUEUES=8" "GPU_MAX_HW_QUEUES=2" "GPU_MAX_HW_QUEUES=8 HIP_FORCE_DEV_KERNARG=1"; do
    if [ "$v" = base ]; then e=""; else e="$v"; fi
    env $e HIFIC_BENCH_GRAPH=0 timeout 200 python bench.py --steps 12 --warmup 4 --no-extras > $O/b.json 2>$O/b.err
    python -c "import json;d=json.loads([l for l in open('$O/b.json') if l.startswith('{')][0]);print('$v rep $rep:', d['ms_per_step'], 'ms', d['value'], 'img/s')" | tee -a $O/ab.log
  done
done
