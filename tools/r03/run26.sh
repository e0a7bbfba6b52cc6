# host rANS coder rewrite + device-side layout: GPU tests of the codec path and the 1 MP compress / decompress timing
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_run26; mkdir -p $O; cd $R
timeout 60 python -m pytest tests/test_gpu_elementwise.py tests/test_gpu_modules.py -k "vectorised_coder or compress" -q -p no:cacheprovider > $O/tests.log 2>&1
echo "codec tests: $(tail -1 $O/tests.log)"; grep -E "^FAILED|^ERROR" $O/tests.log | cut -c1-200
timeout 60 python - <<'PY' 2>&1 | tail -3
import argparse, json, torch, bench
print(json.dumps(bench.codec_leg(argparse.Namespace(dtype="bf16"), torch.device("cuda:0"))))
PY
