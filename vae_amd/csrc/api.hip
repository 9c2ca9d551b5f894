// Library-level entry points of libvae_amd.so (no reference counterpart: they
// describe the ABI itself).
#include <string.h>

#include "common.hpp"
#include "../../include/vae_amd.h"

extern "C" {
int vae_abi_version(void) { return 9; }
// counters + group partials + one partial per block x up to 8 scalars (common.hpp, grid_reduce)
size_t vae_workspace_bytes(void) { return vae::kWorkspaceFloats * sizeof(float); }

int vae_tuning_set(const char* name, int64_t value) {
  if (!name) return VAE_ERR_ARG;
  vae::Knobs& k = vae::knobs();
  if (!strcmp(name, "G")) k.G = (int)value;
  else if (!strcmp(name, "U")) k.U = (int)value;
  else if (!strcmp(name, "MAXGRID")) k.maxgrid = value;
  else if (!strcmp(name, "REDUCEGRID")) k.reducegrid = (int)value;
  else return VAE_ERR_ARG;
  return VAE_OK;
}
}
