R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02_run42; mkdir -p $O
cd $R
b() { env "$@" timeout 300 python bench.py --steps 10 --warmup 3 --no-extras 2>/dev/null | grep metric | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$*:', d['value'], d['ms_per_step'])"; }
b X=0
b HIFIC_FORCE_DIST=1 HIFIC_REDUCE_NONBLOCK=0
b HIFIC_FORCE_DIST=1 HIFIC_REDUCE_NONBLOCK=1
b HIFIC_FORCE_DIST=1 HIFIC_REDUCE_NONBLOCK=0
b HIFIC_FORCE_DIST=1 HIFIC_REDUCE_NONBLOCK=1
HIFIC_FORCE_DIST=1 timeout 600 python -m pytest tests/test_gpu_fullsize.py -q -p no:cacheprovider -k "training_step or side_stream" 2>&1 | tail -2
