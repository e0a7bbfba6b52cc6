R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02_run17; mkdir -p $O
cd $R
for v in 1 0 1 0; do
  HIFIC_NO_WIDE_EPI=$v HIFIC_PROF_DUMP=1 python tools/prof_eval_fwd.py 2> $O/dump_$v.txt | grep fwd
  python tools/prof_layers.py $O/dump_$v.txt 2 > $O/layers_$v.md
done
echo "--- nowide"; head -14 $O/layers_1.md
echo "--- wide"; head -14 $O/layers_0.md
