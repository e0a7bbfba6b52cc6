R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_run10; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_conv.py tests/test_gpu_exact_index.py -q -p no:cacheprovider 2>&1 | tail -4
for ks in 1 0; do
HIFIC_BENCH_ROOFLINE_ONLY=1 HIFIC_KSPLIT=$ks HIFIC_PROF_DUMP=1 timeout 300 python bench.py --steps 4 --warmup 2 > $O/b_ks$ks.json 2> $O/prof_ks$ks.log
echo "KSPLIT=$ks $(python -c "import json; d=json.load(open('$O/b_ks$ks.json')); print(d['ms_per_step'], d['roofline']['gemm_class_ms_per_step'])")"
done
python - <<'PY'
import re,collections
def load(f):
    d=collections.OrderedDict()
    for l in open(f):
        if not l.startswith('HIFIC_PROF'): continue
        t=l.split()
        key=' '.join(t[4:12])
        d.setdefault(key,[]).append(float(t[2]))
    return d
a=load('gpurun_out/r03_run10/prof_ks0.log'); b=load('gpurun_out/r03_run10/prof_ks1.log')
rows=[]
for k in a:
    k2=[x for x in b if x.rsplit(' grid',1)[0]==k.rsplit(' grid',1)[0]]
    if not k2: continue
    ta=sum(a[k])/len(a[k]); tb=sum(b[k2[0]])/len(b[k2[0]])
    if abs(ta-tb)>3: rows.append((ta-tb,k,ta,tb,len(a[k])))
for r in sorted(rows,reverse=True)[:40]: print(f"{r[2]:7.1f} -> {r[3]:7.1f} x{r[4]//4}  {r[1]}")
PY
