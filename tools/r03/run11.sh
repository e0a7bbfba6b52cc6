R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_run11; mkdir -p $O; cd $R
run() { tag=$1; shift; env "$@" HIFIC_BENCH_ROOFLINE_ONLY=1 HIFIC_PROF_DUMP=1 timeout 300 python bench.py --steps 4 --warmup 2 > $O/b_$tag.json 2> $O/prof_$tag.log; echo "$tag $(python -c "import json; d=json.load(open('$O/b_$tag.json')); print(d['ms_per_step'], d['roofline']['gemm_class_ms_per_step'])")"; }
run base HIFIC_X=0
run g400 HIFIC_KSPLIT_MAXGRID=400 HIFIC_KSPLIT_TARGET=1280
run g400b HIFIC_KSPLIT_MAXGRID=400 HIFIC_KSPLIT_TARGET=800
run t1024 HIFIC_KSPLIT_TARGET=1024
python - <<'PY'
import collections
def load(f):
    d=collections.OrderedDict()
    for l in open(f):
        if not l.startswith('HIFIC_PROF'): continue
        t=l.split(); key=' '.join(t[4:12]); d.setdefault(key,[]).append(float(t[2]))
    return {k: sum(v)/len(v) for k,v in d.items()}
a=load('gpurun_out/r03_run11/prof_base.log')
for tag in ('g400','g400b','t1024'):
    b=load(f'gpurun_out/r03_run11/prof_{tag}.log'); print(tag)
    for k in a:
        if k in b and abs(a[k]-b[k])>4: print(f"   {a[k]:7.1f} -> {b[k]:7.1f}  {k}")
PY
