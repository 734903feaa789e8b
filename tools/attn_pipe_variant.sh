#!/bin/bash
# Build an experiment variant of libmmd.so whose attn_pipe_kernel uses another generated stream (timing experiments only):
#   tools/attn_pipe_variant.sh <name> <gen_attn_pipe.py arguments ...>   ->  mm-diffusion_amd/lib/variants/libmmd_<name>.so  (use with MMD_LIB=...)
# e.g.  tools/attn_pipe_variant.sh nomfma --ablate nomfma      tools/attn_pipe_variant.sh gap12 MFMA_GAP=12
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
NAME=$1; shift
cd "$ROOT/mm-diffusion_amd"
SRC=csrc_var_$NAME
rm -rf $SRC && cp -r csrc $SRC
python "$ROOT/tools/gen_attn_pipe.py" "$@" --out $SRC/mmd_attn_pipe_body.inc
mkdir -p lib/variants
OBJ=lib/variants/mmd_attn_$NAME.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -Wno-unused-result -Xclang -target-feature -Xclang -packed-fp32-ops -c $SRC/mmd_attn.hip -o $OBJ 2>&1 | grep -v "not a recognized feature" || true
OTHERS=$(ls lib/mmd_*.o | grep -v "lib/mmd_attn.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o lib/variants/libmmd_$NAME.so $OBJ $OTHERS
rm -rf $OBJ $SRC
echo built lib/variants/libmmd_$NAME.so
