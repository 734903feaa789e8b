#!/bin/bash
# Build (here: hipcc cross-compiles gfx950 without a GPU) or run (on the GPU box) the packed-fp32 reproducer.
#   tools/pk_f32_repro.sh build      -> tools/_pk/pk_repro_packed, tools/_pk/pk_repro_nopk
#   tools/pk_f32_repro.sh run [replays] [reps per graph]   (prints one summary line per binary and per channel count)
set -e
cd "$(dirname "$0")/.."
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=fast -Wno-unused-result -w"
if [ "$1" = build ]; then
  mkdir -p tools/_pk
  /opt/rocm/bin/hipcc $FLAGS tools/pk_f32_repro.hip -o tools/_pk/pk_repro_packed 2>&1 | grep -v "not a recognized feature" || true
  /opt/rocm/bin/hipcc $FLAGS -DPK_REPRO_NOPK -Xclang -target-feature -Xclang -packed-fp32-ops tools/pk_f32_repro.hip -o tools/_pk/pk_repro_nopk 2>&1 | grep -v "not a recognized feature" || true
  ls -la tools/_pk
else
  R=${2:-300}; K=${3:-6}
  for C in 128 256; do
    for b in pk_repro_packed pk_repro_nopk; do
      timeout 300 tools/_pk/$b $R $K $C || true
    done
  done
fi
