#!/bin/bash
# Same-box comparison of the shipped library with another build (default: round 5's final library, built next to it by
#   bash tools/build_ref_lib.sh 92ad164 r05final ):
# graph-replayed UNet forward at batch 8 / batch 1 / the hires shape and the VAE decodes, alternating the two libraries twice.
# Usage (gpurun): bash tools/same_box_r06.sh [other .so] > gpurun_out/r06_same_box_vs_round5.txt
REF=${1:-lightdiffusion_amd/libld_r05final.so}
echo "# same box, alternating: ref = $REF ($(sha256sum $REF | cut -c1-16)), shipped = $(sha256sum lightdiffusion_amd/libld_mi355x.so | cut -c1-16)"
for rep in 1 2; do
  for L in "$REF" ""; do
    tag=$([ -z "$L" ] && echo shipped || echo ref)
    for cmd in "tools/unet_time.py 8 1" "tools/vae_time.py 8 64" "tools/vae_time.py 4 128"; do
      echo -n "[$tag] "; LD_MI355X_LIB=$L python $cmd 2>/dev/null | tr '\n' ' '; echo
    done
    echo -n "[$tag] "; AB_HW=128 LD_MI355X_LIB=$L python tools/unet_time.py 4 2>/dev/null | tr '\n' ' '; echo
  done
done
