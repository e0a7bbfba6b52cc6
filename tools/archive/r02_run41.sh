R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02_run41; mkdir -p $O
cd $R
b() { env "$@" timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-traffic --no-extras | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$*:', d['value'], d['ms_per_step'])"; }
b X=0
b HIFIC_GC_BIGTILE_MIN_GRID=128
b HIFIC_GC_BIGTILE_MIN_GRID=512
b HIFIC_GC_BIGTILE_MIN_GRID=1024
b HIFIC_WG_NOSPLIT=96
b HIFIC_WG_NOSPLIT=256
b X=1
