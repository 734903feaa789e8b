#!/bin/bash
# Build an experiment variant of libmmd.so from ONE re-compiled source + the product objects of the other sources:
#   tools/build_variant.sh <name> <source.hip> "<extra flags>"   ->  mm-diffusion_amd/lib/variants/libmmd_<name>.so   (use with MMD_LIB=...)
set -e
cd "$(dirname "$0")/../mm-diffusion_amd"
NAME=$1; SRC=$2; FLAGS=$3
mkdir -p lib/variants
OBJ=lib/variants/${SRC%.hip}_$NAME.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -Wno-unused-result -Xclang -target-feature -Xclang -packed-fp32-ops $FLAGS -c csrc/$SRC -o $OBJ
OTHERS=$(ls lib/mmd_*.o | grep -v "lib/${SRC%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o lib/variants/libmmd_$NAME.so $OBJ $OTHERS
rm -f $OBJ
echo built lib/variants/libmmd_$NAME.so
