#!/bin/bash
# Developer helper: build lib/libtmpc_<name>.so from an alternative tmpc_kernels.hip (path in $2) for A/B timing.
set -e
cd "$(dirname "$0")/../robust-tracking-mpc-over-lossy-networks_amd/csrc"
mkdir -p /tmp/wk/$1 && cp "$2" /tmp/wk/$1/tmpc_kernels.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC -shared -I. -o ../lib/libtmpc_$1.so -x hip tmpc_api.cpp tmpc_condense.cpp /tmp/wk/$1/tmpc_kernels.hip tmpc_stream.hip tmpc_block.hip tmpc_mc.hip tmpc_lp.hip -mllvm -amdgpu-mfma-vgpr-form=1 ${EXTRA}
