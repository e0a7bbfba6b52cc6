# planner knobs on the split-bf16 (3C-channel) stride-2 Encoder layers, kernel-only timing
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_run20; mkdir -p $O; cd $R
export MOPS=fwd MS=2
for shape in "180 120 256" "360 240 128" "720 480 64" "1440 960 32"; do
  set -- $shape
  for knob in "" "HIFIC_BC32=0" "HIFIC_BM=64" "HIFIC_WSTAGE=0" "HIFIC_GC_BIGTILE_MIN_GRID=1000000" "HIFIC_BC32=0 HIFIC_GC_BIGTILE_MIN_GRID=1000000" "HIFIC_KSPLIT_MAXGRID=600"; do
    env MC=$1 MK=$2 MH=$3 $knob timeout 100 python tools/micro_sp9.py 20 2>/dev/null
  done
done | tee $O/knobs.log
