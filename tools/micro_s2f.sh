# stride-2 FORWARD convs of the GAN cycle, one at a time (MPROF: the conv kernel alone)
for cfg in "16 60 120 256 3 1,0,0,1" "16 120 240 128 3 1,0,0,1" "16 240 480 64 3 1,0,0,1" "16 480 960 32 3 1,0,0,1" \
           "32 64 128 128 4 1,1,1,1" "32 128 256 64 4 1,1,1,1" "32 256 512 32 4 1,1,1,1"; do
  set -- $cfg
  echo -n "N$1 C$2 K$3 H$4 R$5: "
  MPROF=1 MN=$1 MC=$2 MK=$3 MH=$4 MR=$5 MS=2 MPAD=$6 python tools/micro_conv.py ${WHICH:-fwd} 30 2>&1 | tail -1
done
