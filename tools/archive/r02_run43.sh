R=$GRAFT_REPO_ROOT; cd $R
b() { env "$@" timeout 300 python bench.py --steps 10 --warmup 3 --no-extras 2>/dev/null | grep metric | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$*:', d['value'], d['ms_per_step'])"; }
b HIFIC_FORCE_DIST=1 HIFIC_BUCKET_MB=32
b HIFIC_FORCE_DIST=1 HIFIC_BUCKET_MB=128
b HIFIC_FORCE_DIST=1 HIFIC_BUCKET_MB=512
b HIFIC_FORCE_DIST=1 HIFIC_BUCKET_MB=8
