# merged-phase kernel: 16-byte pair stores through LDS (epi_wide 4) vs 4-byte pair stores, parity + same-box A/B
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_run9; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_modules.py tests/test_gpu_exact_generator.py -q -x -p no:cacheprovider > $O/t.log 2>&1; tail -4 $O/t.log
for rep in 1 2; do
  for w in 1 0; do
    HIFIC_MP_WIDE=$w HIFIC_BENCH_ROOFLINE_ONLY=1 timeout 300 python bench.py --steps 10 --warmup 3 --no-traffic --no-parity --no-cpu-baseline > $O/b_w${w}_$rep.json 2> $O/b_w${w}_$rep.err
    python - <<PY
import json
d=json.load(open("$O/b_w${w}_$rep.json"))
pk=d["roofline"]["per_kernel"]
print("MP_WIDE=$w rep $rep:", d["ms_per_step"], "ms;", {k:(v["avg_launch_us"],v["launches_per_step"]) for k,v in pk.items() if "mp" in k})
PY
  done
done
