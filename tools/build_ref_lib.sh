#!/bin/bash
# Build the library of another commit next to the current one, for same-box comparisons (tools/unet_time.py, tools/gemm_time.py):
#   bash tools/build_ref_lib.sh <commit> <name>   ->  lightdiffusion_amd/libld_<name>.so   (git-ignored; travels with gpurun)
set -e
C=${1:-HEAD}; N=${2:-ref}
D=$(mktemp -d)
git archive "$C" lightdiffusion_amd/csrc include | tar -x -C "$D"
make -C "$D/lightdiffusion_amd/csrc" -j8 > /dev/null
cp "$D/lightdiffusion_amd/libld_mi355x.so" "lightdiffusion_amd/libld_$N.so"
rm -rf "$D"
echo "built lightdiffusion_amd/libld_$N.so from $C"
