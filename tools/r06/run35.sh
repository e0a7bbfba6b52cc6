# split tail buckets (HIFIC_BUCKET_TAIL_MB): two-rank tests + rehearsal, then one rank through RCCL with / without the split
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_run35; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_two_ranks.py -q -x -p no:cacheprovider > $O/t.log 2>&1; tail -2 $O/t.log
for rep in 1 2 3; do
  for v in "HIFIC_BUCKET_TAIL_MB=2,32" "HIFIC_BUCKET_TAIL_MB=0"; do
    env $v HIFIC_FORCE_DIST=1 timeout 300 python bench.py --steps 12 --warmup 4 --no-extras > $O/b.json 2>$O/b.err
    python -c "import json;d=json.loads(open('$O/b.json').read());r=d['rccl'];print('$v rep $rep:', d['ms_per_step'], 'ms; exposed', r['exposed_comm_ms'], 'buckets', r['buckets'], 'solo', r['one_rank_same_box_ms_per_step'], [ (x['wire_mbytes'], x['issue_ms']) for x in r['buckets_timeline']['amort']])" | tee -a $O/ab.log
  done
done
