#!/bin/bash
# diagnostic builds of the two largest tile-fused kernels with in-kernel phase stamps (csrc/uir_stamps.h)
cd "$(dirname "$0")/../../../underwater_image_restoration_amd/csrc" || exit 1
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared"
O=../../tools/probes/stamps
/opt/rocm/bin/hipcc $F -DUIR_STAMPS=1 dwln.hip tiled.hip -o $O/libstamps_dwln.so && \
/opt/rocm/bin/hipcc $F -DUIR_STAMPS=1 -fno-slp-vectorize gdfn.hip tiled.hip -o $O/libstamps_gdfn.so && \
for v in 2 3; do
  /opt/rocm/bin/hipcc $F -DUIR_STAMPS=$v -fno-slp-vectorize gdfn.hip tiled.hip -o $O/libfence${v}_gdfn.so
  /opt/rocm/bin/hipcc $F -DUIR_STAMPS=$v dwln.hip tiled.hip -o $O/libfence${v}_dwln.so
done
