#!/bin/bash
# Host AddressSanitizer run of libgims_hip's C++ side (CPU container only; never on the GPU pool: gpurun refuses GPU ASan).
# Device code is compiled normally (-fno-gpu-sanitize); only host code is instrumented.
set -e
cd "$(dirname "$0")/../.."
OUT=${TMPDIR:-/tmp}/gims_asan
mkdir -p "$OUT"
SRCS=$(ls gims_amd/csrc/*.hip)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -fsanitize=address -fno-gpu-sanitize -fno-omit-frame-pointer \
    -mllvm -amdgpu-mfma-vgpr-form $SRCS tools/asan/host_paths.cpp -o "$OUT/host_paths"
ASAN_OPTIONS=detect_leaks=1:abort_on_error=1:protect_shadow_gap=0 "$OUT/host_paths"
