#!/bin/bash
# Build libposeadv_hip.so (bf16 storage) and libposeadv_hip_fp16.so (IEEE-half storage, -DPA_FP16) for gfx950 (MI355X) in-tree.
# hipcc cross-compiles without a GPU.  PA_TUNING=1: compile the A/B environment switches in (tools/sweep_env.sh); the default
# (release) build reads no environment.
set -e
cd "$(dirname "$0")"
SRCS="conv_igemm conv3x3_tile conv1x1_tile conv_wgrad conv_wgrad_tile elementwise pose_ops crop_warp net asn api"
build_variant() {      # $1 = object directory, $2 = extra flags, $3 = library name
  local dir=$1 flags=$2 lib=$3
  mkdir -p $dir
  if [ -n "$PA_TUNING" ]; then flags="$flags -DPA_TUNING ${PA_EXTRA:-}"; fi
  if [ "$(cat $dir/.flags 2>/dev/null)" != "$flags" ]; then rm -f $dir/*.o; echo "$flags" > $dir/.flags; fi
  local objs=""
  for f in $SRCS; do
    local o=$dir/$f.o
    local stale=0                 # an object is stale when its source or ANY header is newer (bn_fin.h comes in through kernels.h)
    [ -f $o ] && [ ! $f.hip -nt $o ] || stale=1
    for h in *.h ../../include/poseadv.h; do [ $h -nt $o ] && stale=1; done
    if [ $stale = 1 ]; then
      hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $flags -c $f.hip -o $o &
    fi
    objs="$objs $o"
  done
  wait
  hipcc --offload-arch=gfx950 -shared -fPIC $objs -o ../$lib
  echo "built $(cd .. && pwd)/$lib"
}
# the two variants side by side (the longest translation unit of one overlaps the short ones of the other)
build_variant ../build "" libposeadv_hip.so &
P1=$!
if [ "$PA_ONLY" = "bf16" ]; then        # (tuning trees: sweeps of the bf16 benchmark only)
  wait $P1; exit $?
fi
build_variant ../build_fp16 "-DPA_FP16" libposeadv_hip_fp16.so &
P2=$!
wait $P1; R1=$?
wait $P2; R2=$?
[ $R1 -eq 0 ] || echo "build.sh: the bf16 variant (libposeadv_hip.so) FAILED (rc $R1)" >&2
[ $R2 -eq 0 ] || echo "build.sh: the fp16 variant (libposeadv_hip_fp16.so) FAILED (rc $R2)" >&2
[ $R1 -eq 0 ] && [ $R2 -eq 0 ]
