R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_run9; mkdir -p $O; cd $R
for ag in 2 3; do HIFIC_SP9_AG=$ag timeout 600 python -m pytest tests/test_gpu_conv.py -q -p no:cacheprovider -k "R_3x3_960 or RFX or E6 or L3" 2>&1 | tail -1; done
for ag in 0 1 2 3; do
HIFIC_SP9_AG=$ag HIFIC_SIDE_WGRAD=0 HIFIC_BRANCH_STREAMS=0 HIFIC_PROF_DUMP=1 timeout 300 python bench.py --steps 2 --warmup 2 --no-extras > /dev/null 2> $O/prof_ag$ag.log
echo "AG=$ag"; grep "sp9_kernel<2,2" $O/prof_ag$ag.log | grep "K960 C960" | awk '{n[$2]++; s[$2]+=$3} END {for (k in n) print k, n[k], s[k]/n[k]}'
done
for ag in 3 0; do
  HIFIC_SP9_AG=$ag timeout 300 python bench.py --steps 10 --warmup 3 --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ag=$ag', d['value'], d['ms_per_step'])"
done
