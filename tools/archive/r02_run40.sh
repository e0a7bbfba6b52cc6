R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02_run40; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_conv.py tests/test_gpu_modules.py -q -p no:cacheprovider -x > $O/tests.log 2>&1; tail -2 $O/tests.log
for v in 0 1; do echo "fwd 960x960 DS=$v: $(HIFIC_SP9_DS=$v timeout 120 python tools/micro_conv.py fwd 50 2>&1 | tail -1)"; done
for v in 0 1 0 1; do
  HIFIC_SP9_DS=$v timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-traffic --no-extras | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('DS=$v:', d['value'], d['ms_per_step'])"
done
