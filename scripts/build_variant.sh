#!/bin/bash
# Tuning aid: build a variant of the library with extra -D flags into vae_amd/libvae_amd_<tag>.so
#   scripts/build_variant.sh allnt -DVAE_NT_STORE_MIN_BYTES=0
set -e
tag=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
obj=/tmp/vae_variant_$tag; mkdir -p $obj
for f in $root/vae_amd/csrc/*.hip; do
  b=$(basename $f .hip)
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-pass-failed -fno-gpu-rdc "$@" -c $f -o $obj/$b.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/vae_amd/libvae_amd_$tag.so $obj/*.o
echo built $root/vae_amd/libvae_amd_$tag.so
