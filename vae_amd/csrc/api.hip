// Library-level entry points of libvae_amd.so (no reference counterpart: they
// describe the ABI itself).
#include "common.hpp"
#include "../../include/vae_amd.h"

extern "C" {
int vae_abi_version(void) { return 6; }
// counters + group partials + one partial per block x up to 8 scalars (common.hpp, grid_reduce)
size_t vae_workspace_bytes(void) { return vae::kWorkspaceFloats * sizeof(float); }
}
