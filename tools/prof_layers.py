"""Aggregate the per-launch dump of the in-library profiler (HIFIC_PROF_DUMP=1 python bench.py ... 2> log)."""
import sys, collections
agg = collections.OrderedDict()
for line in open(sys.argv[1]):
    if not line.startswith("HIFIC_PROF "):
        continue
    _, kind, us, flops, tag = line.rstrip("\n").split(" ", 4)
    a = agg.setdefault(tag, [0, 0.0, 0.0])
    a[0] += 1; a[1] += float(us); a[2] += float(flops)
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
tot = sum(a[1] for a in agg.values())
print(f"| launches/step | us/launch | ms/step | TFLOP/s | shape |\n|---|---|---|---|---|")
for tag, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"| {a[0]/steps:.0f} | {a[1]/a[0]:.1f} | {a[1]/steps/1e3:.3f} | {a[2]/a[1]/1e6:.1f} | {tag} |")
print(f"total {tot/steps/1e3:.3f} ms/step")
