# A operands by LDS-DMA into a wave-private ring (HIFIC_SP9_AL=1 on the four-quarter form): parity + kernel-only timing
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_run23; mkdir -p $O; cd $R
HIFIC_SP9_AL=1 timeout 300 python -m pytest tests/test_gpu_conv.py tests/test_gpu_fullsize.py -k "bf16 and (test_conv2d_fwd_bwd or resblock_960)" -q -p no:cacheprovider > $O/tests_al.log 2>&1
echo "AL: $(tail -1 $O/tests_al.log)"; grep -E "^FAILED|^ERROR" $O/tests_al.log | cut -c1-200
for rep in 1 2; do
  timeout 120 python tools/micro_sp9.py 40 2>/dev/null
  HIFIC_SP9_AL=1 timeout 120 python tools/micro_sp9.py 40 2>/dev/null
done | tee $O/al.log
