#!/bin/bash
# scratch build with cycle stamps of workgroup 0 (-DAC_SPLIT_TIMING) -> variants/libclk.so (git-ignored; read with tools/diag/clk_pair.py)
set -e
cd "$(dirname "$0")/.."
mkdir -p variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-hip-fp32-correctly-rounded-divide-sqrt -fgpu-flush-denormals-to-zero -fno-slp-vectorize \
  -DAC_SPLIT_TIMING -fPIC -shared -o variants/libclk.so aircombat-selfplay_amd/csrc/aircombat.hip
