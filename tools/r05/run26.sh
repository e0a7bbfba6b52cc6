R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_run26; mkdir -p $O; cd $R
timeout 100 python tools/r05/micro_d1.py 2>&1 | grep d1_ctx | tee $O/d1.log
timeout 900 python -m pytest tests/test_gpu_modules.py -m gpu -x -q -p no:cacheprovider -k "context_gradient or discriminator" > $O/tests.log 2>&1; tail -2 $O/tests.log
