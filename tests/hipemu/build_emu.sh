#!/bin/bash
# TEST INFRASTRUCTURE: compile the real kernel sources for the HOST against tests/hipemu/hip/hip_runtime.h
set -e
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
OUT="${1:-$ROOT/tests/hipemu/libsegsde_emu.so}"
/opt/rocm/lib/llvm/bin/clang++ -x c++ -std=c++20 -O2 -fPIC -shared -ffp-contract=off -mavx2 -mfma \
  -I "$ROOT/tests/hipemu" -Wno-unused-value -Wno-psabi \
  "$ROOT"/improving_segmentation_with_selfsupervised_depth_amd/csrc/*.hip -o "$OUT"
echo "$OUT"
