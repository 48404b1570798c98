#!/bin/bash
# Builds the GEMM probe (scripts/gemm_probe.hip) and, with a commit argument, a reference build of libwjhip from that
# commit's sources as whisperjav_amd/csrc/libwjhip_ref.so (git-ignored; travels to the GPU box with the working tree).
#   scripts/build_probe.sh [REF_COMMIT]
set -euo pipefail
cd "$(dirname "$0")/.."
CSRC=whisperjav_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wno-unused-result"
if [ $# -ge 1 ]; then
  REF=$CSRC/build/ref_src
  rm -rf "$REF"; mkdir -p "$REF"
  git archive "$1" $CSRC include | tar -x -C "$REF"
  objs=()
  for f in "$REF"/$CSRC/*.hip; do
    o="$REF/$(basename "${f%.hip}").o"
    extra=""; [ "$(basename "$f")" = attention.hip ] && extra="-mllvm -amdgpu-mfma-vgpr-form"
    hipcc $FLAGS $extra -c "$f" -o "$o" &
    objs+=("$o")
  done
  wait
  hipcc --offload-arch=gfx950 -shared -fPIC -o $CSRC/libwjhip_ref.so "${objs[@]}" -ldl
  rm -rf "$REF"
  echo "built $CSRC/libwjhip_ref.so from $1"
fi
hipcc --offload-arch=gfx950 -O2 scripts/gemm_probe.hip -o $CSRC/gemm_probe -Iinclude -ldl
echo "built $CSRC/gemm_probe"
