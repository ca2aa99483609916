#!/bin/bash
# Device ISA of one translation unit of libsdpgpu.so (same flags as build.py):  tools/isa.sh sdpgpu_window.hip > /tmp/window.s
# Resource lines (.vgpr_count, .sgpr_count, spills, LDS) are in the .amdgpu_metadata block at the end of the file.
set -e
cd "$(dirname "$0")/../stochastic-inventory_amd/csrc"
exec /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math --cuda-device-only -S -o - "$1"
