#!/bin/bash
# TEST INFRASTRUCTURE: compile the real kernel sources for the HOST against tests/hipemu/hip/hip_runtime.h
# (one object per source, in parallel, only what changed; then one link)
set -e
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
OUT="${1:-$ROOT/tests/hipemu/libsegsde_emu.so}"
OBJ="$ROOT/build/emu_obj"
CSRC="$ROOT/improving_segmentation_with_selfsupervised_depth_amd/csrc"
mkdir -p "$OBJ"
CXX=/opt/rocm/lib/llvm/bin/clang++
FLAGS="-x c++ -std=c++20 -O2 -fPIC -ffp-contract=off -mavx2 -mfma -pthread -ftls-model=initial-exec -I $ROOT/tests/hipemu -Wno-unused-value -Wno-psabi"
newest_hdr=$(ls -t "$CSRC"/*.h "$ROOT"/include/*.h "$ROOT"/tests/hipemu/hip/*.h "$0" | head -1)
todo=()
for s in "$CSRC"/*.hip; do
  o="$OBJ/$(basename "$s").o"
  if [ ! -e "$o" ] || [ "$s" -nt "$o" ] || [ "$newest_hdr" -nt "$o" ]; then todo+=("$s"); fi
done
if [ ${#todo[@]} -gt 0 ]; then
  printf '%s\n' "${todo[@]}" | xargs -P 8 -I{} sh -c "$CXX $FLAGS -c {} -o $OBJ/\$(basename {}).o"
fi
$CXX -shared -pthread "$OBJ"/*.hip.o -o "$OUT"
echo "$OUT"
