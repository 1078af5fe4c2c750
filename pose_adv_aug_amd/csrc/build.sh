#!/bin/bash
# Build libposeadv_hip.so for gfx950 (MI355X) in-tree.  hipcc cross-compiles without a GPU.
set -e
cd "$(dirname "$0")"
mkdir -p ../build
# PA_TUNING=1: compile the A/B environment switches in (tools/sweep_env.sh); the default (release) build has none
FLAGS=""
if [ -n "$PA_TUNING" ]; then FLAGS="-DPA_TUNING"; fi
if [ "$(cat ../build/.flags 2>/dev/null)" != "$FLAGS" ]; then rm -f ../build/*.o; echo "$FLAGS" > ../build/.flags; fi
OBJS=""
for f in conv_igemm conv3x3_tile conv1x1_tile conv_wgrad conv_wgrad_tile elementwise pose_ops crop_warp net asn api; do
  if [ ! -f ../build/$f.o ] || [ $f.hip -nt ../build/$f.o ] || [ common.h -nt ../build/$f.o ] || [ kernels.h -nt ../build/$f.o ] || [ conv_epilogue.h -nt ../build/$f.o ] || [ net.h -nt ../build/$f.o ] || [ pose_ops.h -nt ../build/$f.o ] || [ ../../include/poseadv.h -nt ../build/$f.o ]; then
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $FLAGS -c $f.hip -o ../build/$f.o &
  fi
  OBJS="$OBJS ../build/$f.o"
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o ../libposeadv_hip.so
echo "built $(cd .. && pwd)/libposeadv_hip.so"
