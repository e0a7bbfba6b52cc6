R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_run11; mkdir -p $O; cd $R
HIFIC_PROF_DUMP=1 HIFIC_BENCH_ROOFLINE_ONLY=1 timeout 300 python bench.py --steps 6 --warmup 3 --no-traffic --no-parity --no-cpu-baseline > $O/b.json 2> $O/b.err
grep HIFIC_PROF $O/b.err | wc -l
python - <<PY
import collections
rows=collections.defaultdict(list)
for l in open("$O/b.err"):
    if not l.startswith("HIFIC_PROF "): continue
    p=l.split(None,4)
    name=p[1]; 
    # name may contain spaces ("narrow"): re-split from the right: us flops tag...
    toks=l[len("HIFIC_PROF "):].rstrip().split(" ")
    # find first token that parses as float -> us
    i=0
    while True:
        try: float(toks[i]); break
        except: i+=1
    name=" ".join(toks[:i]); us=float(toks[i]); fl=float(toks[i+1]); tag=" ".join(toks[i+2:])
    rows[(name,tag)].append((us,fl))
tot=sum(sum(u for u,_ in v) for v in rows.values())
ncyc=4.0
out=[]
for (name,tag),v in rows.items():
    us=sum(u for u,_ in v)/len(v); n=len(v)/ncyc
    out.append((us*n,us,n,v[0][1],name,tag))
out.sort(reverse=True)
print("GEMM-class per cycle: %.2f ms" % (tot/ncyc/1e3))
for t,us,n,fl,name,tag in out[:70]:
    print("%7.1f us/cycle  %6.1f us x%4.1f  %5.0f TF/s  %-34s %s" % (t,us,n,fl/us/1e6,name,tag))
PY
