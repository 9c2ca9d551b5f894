#!/bin/bash
# A/B of library variants (scripts/build_variant.sh) on the Gumbel forward, philox y-only (the training path).
for tag in "" $@; do
  lib=vae_amd/libvae_amd${tag:+_$tag}.so
  echo "=== ${tag:-base}"
  VAE_AMD_LIB=$PWD/$lib BENCH_COLD_SAMPLES=15 timeout -k 10 120 python -c "
import sys; sys.path.insert(0,'scripts')
import bench_configs as bc
bc.gumbel(65536,256,philox=True)
bc.gauss(65536,128,philox=True)
" 2>&1 | grep -E "y only|gumbel_fwd  |gauss_reparam_kl_fwd .*philox  " || exit 1
done
