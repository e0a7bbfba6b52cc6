# round 6, call 2: the exact Generator chain - kernel tests, module tests, the oracle-pinned full-size gradients, cycle A/B
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_run2; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_exact_generator.py -q -s -p no:cacheprovider > $O/t_exact_gen.log 2>&1; tail -15 $O/t_exact_gen.log
timeout 600 python -m pytest tests/test_gpu_golden.py -q -s -p no:cacheprovider -k "exact_training or exact_reconstruction" > $O/t_golden.log 2>&1; tail -8 $O/t_golden.log
timeout 300 python tools/r06/exact_ab.py 8 > $O/exact_ab.log 2>&1; grep exact_ab $O/exact_ab.log
timeout 900 python -m pytest tests/test_gpu_fullsize_backward.py -q -s -p no:cacheprovider -k "bf16_modes" > $O/t_fullsize.log 2>&1; tail -30 $O/t_fullsize.log
echo done
