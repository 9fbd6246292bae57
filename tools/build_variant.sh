#!/bin/bash
# tools/build_variant.sh <name> <src.hip to recompile> [extra -D flags...]
# Builds ablate/libsea_<name>.so = the product library with ONE translation unit recompiled with
# extra macros (timing-only diagnostic variants; results may be wrong by construction).
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
C=$ROOT/speech_enhancement_amd/csrc
name=$1; src=$2; shift 2
mkdir -p $ROOT/ablate
base=$(basename $src .hip)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 -fno-gpu-flush-denormals-to-zero -fno-slp-vectorize \
    -I$C "$@" -c $src -o /tmp/variant_$name.o
objs=""
for o in capi hostpipe mapping ns_kernel ns_pipe_kernel ns_pipe6_kernel ns_pipe2_kernel ns_wave_kernel ns16k_kernel ns16k_pipe_kernel cc_kernel resynth_kernel irm_kernel; do
  if [ "$o" = "$base" ] || [ "$o" = "${VARIANT_REPLACES:-}" ]; then objs="$objs /tmp/variant_$name.o"; else objs="$objs $C/$o.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o $ROOT/ablate/libsea_$name.so $objs $C/sea_tables.o
echo built ablate/libsea_$name.so
