#!/bin/bash
# tools/build_variant.sh NAME "-DFLAG=..."  ->  tools/build/libvitssl_NAME.so  (kernel A/B builds; use with VITSSL_LIB=...)
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/tools/build"; mkdir -p "$OUT/$1"
CS="${CSRC_DIR:-$ROOT/vit-ssl_amd/csrc}"   # CSRC_DIR: sources of another commit (git archive <rev> vit-ssl_amd/csrc include | tar -x -C <dir>)
objs=()
for f in error.cpp gemm_nt.hip gemm_tn.hip layernorm.hip attention.hip elementwise.hip dino.hip augment.hip fp8.hip; do
  extra=""; [ "$f" = augment.hip ] && extra="-ffp-contract=off"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result $2 $extra -x hip -c "$CS/$f" -o "$OUT/$1/$f.o" &
  objs+=("$OUT/$1/$f.o")
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/libvitssl_$1.so" "${objs[@]}"
echo "built $OUT/libvitssl_$1.so"
