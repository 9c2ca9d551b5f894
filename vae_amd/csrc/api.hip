// Library-level entry points of libvae_amd.so (no reference counterpart: they
// describe the ABI itself).
#include <string.h>

#include "common.hpp"
#include "../../include/vae_amd.h"

namespace {
// Philox call offsets for draws recorded into HIP graphs: one counter per GPU (a file-scope __device__ word: every
// device the library runs on has its own), advanced by a one-thread node of the graph itself.
__device__ unsigned long long g_graph_calls;
__global__ void rng_next_kernel(unsigned long long* cell, unsigned long long n) {
  // one atomic add (a vector-memory atomic at L2): graphs replaying on two streams of a GPU never hand out the same
  // offsets, and the read does not go through the scalar cache
  const unsigned long long c = __hip_atomic_fetch_add(&g_graph_calls, n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  *cell = VAE_GRAPH_OFFSET_BASE + c + 1ull;
}
__global__ void rng_reset_kernel() { g_graph_calls = 0ull; }
}  // namespace

extern "C" {
int vae_abi_version(void) { return 11; }

int vae_rng_next(uint64_t* cell, int64_t n, vae_stream_t stream) {
  if (!cell || n < 1) return VAE_ERR_ARG;
  hipLaunchKernelGGL(rng_next_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (unsigned long long*)cell,
                     (unsigned long long)n);
  return hipGetLastError() == hipSuccess ? VAE_OK : VAE_ERR_LAUNCH;
}
int vae_rng_reset(vae_stream_t stream) {
  hipLaunchKernelGGL(rng_reset_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream);
  return hipGetLastError() == hipSuccess ? VAE_OK : VAE_ERR_LAUNCH;
}
// counters + group partials + one partial per block x up to 8 scalars (common.hpp, grid_reduce)
size_t vae_workspace_bytes(void) { return vae::kWorkspaceFloatsFull * sizeof(float); }

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
