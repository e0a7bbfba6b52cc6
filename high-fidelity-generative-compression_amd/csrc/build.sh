#!/bin/bash
# Build libhific_hip.so for gfx950 (cross-compiles without a GPU).  Output lands next to the package.
set -e
cd "$(dirname "$0")"
OUT=../libhific_hip.so
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -Wno-unused-value"
OBJS=""
for f in gconv elementwise norm entropy lpips capi; do
  if [ ! -f $f.o ] || [ $f.hip -nt $f.o ] || [ common.h -nt $f.o ] || [ gconv.h -nt $f.o ]; then
    hipcc $FLAGS -c $f.hip -o $f.o &
  fi
  OBJS="$OBJS $f.o"
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o $OUT
# host-side (CPU) table construction for the EVALUATION path: plain g++, no device code
g++ -O2 -std=c++17 -fPIC -shared -fno-fast-math host_tables.cpp host_rans.cpp -o ../libhific_host.so
echo "built $OUT"
