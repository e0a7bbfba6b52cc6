# four-quarter sp9 forms (HIFIC_SP9_W4=1: 8 waves x 64x128 slabs, =2: 4 waves x 128x128 slabs) and the 16-byte-per-lane
# ChannelNorm kernels (HIFIC_CN_{FWD,BWD}_V8): parity first, then kernel-level and whole-cycle A/B on one box
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_run17; mkdir -p $O; cd $R
for w in 1 2; do
  HIFIC_SP9_W4=$w timeout 300 python -m pytest tests/test_gpu_conv.py tests/test_gpu_fullsize.py -k "bf16 and (test_conv2d_fwd_bwd or resblock_960)" -q -p no:cacheprovider > $O/tests_w4_$w.log 2>&1
  echo "W4=$w: $(tail -1 $O/tests_w4_$w.log)"; grep -E "^FAILED|^ERROR" $O/tests_w4_$w.log | cut -c1-200
done
HIFIC_CN_FWD_V8=1 HIFIC_CN_BWD_V8=1 timeout 300 python -m pytest tests/test_gpu_elementwise.py -k channelnorm -q -p no:cacheprovider > $O/tests_cn.log 2>&1
echo "CN v8: $(tail -1 $O/tests_cn.log)"; grep -E "^FAILED|^ERROR" $O/tests_cn.log | cut -c1-200
for v in 0 1; do HIFIC_CN_FWD_V8=$v HIFIC_CN_BWD_V8=$v timeout 120 python tools/micro_cn.py 2>/dev/null | sed "s/^/v8=$v /"; done
for v in 0 1; do HIFIC_CN_FWD_V8=$v HIFIC_CN_BWD_V8=0 timeout 120 python tools/micro_cn.py 2>/dev/null | sed "s/^/fwd-only v8=$v /"; done
ab() {   # name, env...
  name=$1; shift
  env "$@" HIFIC_BENCH_ROOFLINE_ONLY=1 timeout 300 python bench.py --steps 8 --warmup 3 2>$O/bench_$name.err > $O/bench_$name.json
  python - "$name" $O/bench_$name.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    pk = d["roofline"]["per_kernel"]
    sp = {k: (round(v["avg_launch_us"], 1), round(v["tflops"])) for k, v in pk.items() if "sp9_kernel<2,2" in k or "sp9_kernel<2,4" in k or "sp9_kernel<4,4" in k}
    print(sys.argv[1], d["value"], d["ms_per_step"], "gemm_ms", d["roofline"]["gemm_class_ms_per_step"], sp)
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
ab base HIFIC_SP9_W4=0
ab w4_1 HIFIC_SP9_W4=1
ab w4_2 HIFIC_SP9_W4=2
ab w4_1_v8 HIFIC_SP9_W4=1 HIFIC_CN_FWD_V8=1 HIFIC_CN_BWD_V8=1
ab base2 HIFIC_SP9_W4=0
