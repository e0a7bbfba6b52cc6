# wide patch staging (HIFIC_SP9_WP=1 on top of HIFIC_SP9_W4=1): parity, kernel-only timing, whole-cycle A/B
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_run21; mkdir -p $O; cd $R
HIFIC_SP9_W4=1 HIFIC_SP9_WP=1 timeout 300 python -m pytest tests/test_gpu_conv.py tests/test_gpu_fullsize.py -k "bf16 and (test_conv2d_fwd_bwd or resblock_960)" -q -p no:cacheprovider > $O/tests_wp.log 2>&1
echo "WP: $(tail -1 $O/tests_wp.log)"; grep -E "^FAILED|^ERROR" $O/tests_wp.log | cut -c1-200
for rep in 1 2; do
  HIFIC_SP9_W4=1 MOPS=fwd timeout 120 python tools/micro_sp9.py 40 2>/dev/null
  HIFIC_SP9_W4=1 HIFIC_SP9_WP=1 MOPS=fwd timeout 120 python tools/micro_sp9.py 40 2>/dev/null
done
ab() {
  name=$1; shift
  env "$@" HIFIC_BENCH_ROOFLINE_ONLY=1 timeout 300 python bench.py --steps 8 --warmup 3 2>$O/bench_$name.err > $O/bench_$name.json
  python - "$name" $O/bench_$name.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    pk = d["roofline"]["per_kernel"]
    sp = {k: (round(v["avg_launch_us"], 1), round(v["tflops"])) for k, v in pk.items() if "sp9_kernel<2,2" in k or "sp9_kernel<2,4" in k}
    print(sys.argv[1], d["value"], d["ms_per_step"], "gemm_ms", d["roofline"]["gemm_class_ms_per_step"], sp)
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
V8="HIFIC_CN_FWD_V8=1 HIFIC_CN_BWD_V8=1"
ab w4_v8 HIFIC_SP9_W4=1 $V8
ab w4_wp_v8 HIFIC_SP9_W4=1 HIFIC_SP9_WP=1 $V8
ab w4_v8_b HIFIC_SP9_W4=1 $V8
ab w4_wp_v8_b HIFIC_SP9_W4=1 HIFIC_SP9_WP=1 $V8
