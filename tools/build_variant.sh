#!/bin/bash
# scratch build of the library with extra -D flags -> variants/lib<name>.so (git-ignored): bash tools/build_variant.sh <name> -DFOO=1 ...
set -e
cd "$(dirname "$0")/.."
mkdir -p variants
name=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-hip-fp32-correctly-rounded-divide-sqrt -fgpu-flush-denormals-to-zero -fno-slp-vectorize \
  "$@" -fPIC -shared -o variants/lib$name.so aircombat-selfplay_amd/csrc/aircombat.hip
