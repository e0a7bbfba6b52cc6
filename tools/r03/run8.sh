R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_run8; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_conv.py -q -p no:cacheprovider -k "R_3x3_960 or RFX or E6 or pack_cache or L3 or A1" 2>&1 | tail -4
for ag in 1 0; do
  HIFIC_SP9_AG=$ag timeout 300 python bench.py --steps 10 --warmup 3 --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ag=$ag', d['value'], d['ms_per_step'])"
done
for ag in 1 0; do
HIFIC_SP9_AG=$ag HIFIC_SIDE_WGRAD=0 HIFIC_BRANCH_STREAMS=0 HIFIC_PROF_DUMP=1 timeout 300 python bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-traffic --no-parity > $O/prof_ag$ag.json 2> $O/prof_ag$ag.log
grep "sp9_kernel<2,2" $O/prof_ag$ag.log | awk '{print $2, $3, $6, $7}' | sort | uniq -c | sort -rn | head -8
done
