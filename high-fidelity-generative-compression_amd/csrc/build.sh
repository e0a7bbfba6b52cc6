#!/bin/bash
# Build libhific_hip.so for gfx950 (cross-compiles without a GPU).  Output lands next to the package.
set -e
cd "$(dirname "$0")"
# A/B builds: EXTRA="-DGC_TOFF_EARLY=0" OUT=../libhific_hip_ab.so OBJDIR=/tmp/ab bash build.sh ; run with HIFIC_LIB_PATH=<that .so>
OUT=${OUT:-../libhific_hip.so}
OBJDIR=${OBJDIR:-.}
mkdir -p $OBJDIR
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -Wno-unused-value $EXTRA"
SRCS="gconv gconv_mpvc gconv_sp9 gconv_pack gconv_wgrad gconv_wgrad_nat gconv_wgrad_c3 gconv_pl gconv_wr elementwise norm entropy lpips augment capi"
OBJS=""
PIDS=""
NAMES=""
for f in $SRCS; do
  if [ ! -f $OBJDIR/$f.o ] || [ $f.hip -nt $OBJDIR/$f.o ] || [ common.h -nt $OBJDIR/$f.o ] || [ gconv.h -nt $OBJDIR/$f.o ] || [ gconv_dev.h -nt $OBJDIR/$f.o ] || [ gconv_stage.h -nt $OBJDIR/$f.o ]; then
    rm -f $OBJDIR/$f.o                            # a failed compile must never leave a stale object to link
    hipcc $FLAGS -c $f.hip -o $OBJDIR/$f.o &
    PIDS="$PIDS $!"
    NAMES="$NAMES $f"
  fi
  OBJS="$OBJS $OBJDIR/$f.o"
done
# a bare `wait` always returns 0: collect every compile's status
i=0
for pid in $PIDS; do
  i=$((i + 1))
  if ! wait $pid; then
    echo "hipcc failed on $(echo $NAMES | cut -d' ' -f$i).hip" >&2
    exit 1
  fi
done
hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o $OUT
# measurement-only twin (bench.py `roofline.practical_peak`): the same library with the residual-trunk kernel's memory instructions
# compiled out (-DSP9_ABL=15: no patch loads, no weight loads, no LDS fragment reads, no chunk barrier - WRONG RESULTS by
# construction), i.e. the bare MFMA stream of that tile on resident operands.  Never loaded by the product (lib.py loads $OUT).
if [ -z "$EXTRA" ] && [ "$OUT" = "../libhific_hip.so" ]; then
  if [ ! -f $OBJDIR/gconv_sp9_abl.o ] || [ gconv_sp9.hip -nt $OBJDIR/gconv_sp9_abl.o ] || [ gconv_stage.h -nt $OBJDIR/gconv_sp9_abl.o ] || [ gconv_dev.h -nt $OBJDIR/gconv_sp9_abl.o ] || [ gconv.h -nt $OBJDIR/gconv_sp9_abl.o ]; then
    hipcc $FLAGS -DSP9_ABL=15 -c gconv_sp9.hip -o $OBJDIR/gconv_sp9_abl.o
  fi
  hipcc --offload-arch=gfx950 -shared -fPIC $(echo $OBJS | sed "s#$OBJDIR/gconv_sp9.o#$OBJDIR/gconv_sp9_abl.o#") -o ../libhific_hip_mfma_only.so
fi
# host-side (CPU) table construction + rANS coder for the EVALUATION path: plain g++, no device code
g++ -O2 -std=c++17 -fPIC -shared -fno-fast-math host_tables.cpp host_rans.cpp -o ../libhific_host.so
echo "built $OUT"
