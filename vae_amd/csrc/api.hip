// Library-level entry points of libvae_amd.so (no reference counterpart: they
// describe the ABI itself).
#include "common.hpp"
#include "../../include/vae_amd.h"

extern "C" {
int vae_abi_version(void) { return 1; }
// one partial per block (<= kMaxGrid) x up to 8 scalars per kernel
size_t vae_workspace_bytes(void) { return (size_t)vae::kMaxGrid * 8 * sizeof(float); }
}
