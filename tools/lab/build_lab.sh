#!/bin/bash
# builds tools/lab/bin/fused_lab from the two march kernel files (no torch, no libxpt_hip.so)
set -e
cd "$(dirname "$0")/../.."
mkdir -p tools/lab/bin
hipcc -O3 -fno-slp-vectorize --offload-arch=gfx950 -std=c++17 -Wno-unused-function $EXTRA -o tools/lab/bin/fused_lab tools/lab/fused_lab.hip \
      xpt_mde_2021_amd/csrc/xpt_fused.hip xpt_mde_2021_amd/csrc/xpt_march.hip
