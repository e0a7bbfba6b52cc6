R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_run12; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_conv.py -q -p no:cacheprovider 2>&1 | tail -2
run() { tag=$1; shift; env "$@" HIFIC_BENCH_ROOFLINE_ONLY=1 HIFIC_PROF_DUMP=1 timeout 300 python bench.py --steps 4 --warmup 2 > $O/b_$tag.json 2> $O/prof_$tag.log; echo "$tag $(python -c "import json; d=json.load(open('$O/b_$tag.json')); print(d['ms_per_step'], d['roofline']['gemm_class_ms_per_step'], {k: round(v['ms_per_step'],2) for k,v in d['roofline']['per_kernel'].items()})")"; }
run old HIFIC_LIB_PATH=$R/tools/ab/libhific_old.so
run new HIFIC_X=0
run ag4 HIFIC_SP9_AG=4
run old2 HIFIC_LIB_PATH=$R/tools/ab/libhific_old.so
run new2 HIFIC_X=0
