// Shared device helpers for the vae_amd HIP kernels (gfx950 / CDNA4, wave64).
//
// Every kernel on this path is a streaming row kernel: a "row" is one sample's
// feature vector (L latents, K categories or D pixels).  A row is owned by a
// group of G consecutive lanes (G a power of two, 1..64, picked on the host from
// the row length), each lane moves 16 bytes per access (4 x f32 or 8 x bf16) so
// a wave issues 1 KiB coalesced requests, row reductions are wave64 shuffles
// inside the group, and the few batch-level scalars (means) leave the kernel as
// one partial per block that a one-block finalize kernel sums in a fixed order
// (deterministic: no float atomics).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vae {

constexpr int kBlock = 256;     // 4 waves per workgroup
constexpr int kMaxGrid = 2048;  // 256 CUs x 8 resident workgroups; rows beyond are grid-strided

struct bf16_t {
  uint16_t bits;
};

__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v.bits) << 16); }

template <typename T>
__device__ __forceinline__ T from_f32(float v);
template <>
__device__ __forceinline__ float from_f32<float>(float v) {
  return v;
}
template <>
__device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) {
  // plain cast: hipcc emits v_cvt_pk_bf16_f32 (round-nearest-even, NaN stays NaN)
  __bf16 h = static_cast<__bf16>(v);
  bf16_t r;
  r.bits = __builtin_bit_cast(uint16_t, h);
  return r;
}

// 16-byte packets: the widest per-lane access (global_load/store_dwordx4).
template <typename T, int V>
struct alignas(sizeof(T) * V) Pack {
  T v[V];
};

template <typename T>
struct VecOf {
  static constexpr int n = 16 / sizeof(T);
};

template <typename T, int V>
__device__ __forceinline__ void load_vec(const T* __restrict__ p, float (&out)[V]) {
  if constexpr (V == 1) {
    out[0] = to_f32(p[0]);
  } else {
    Pack<T, V> raw = *reinterpret_cast<const Pack<T, V>*>(p);
#pragma unroll
    for (int i = 0; i < V; ++i) out[i] = to_f32(raw.v[i]);
  }
}

template <typename T, int V>
__device__ __forceinline__ void store_vec(T* __restrict__ p, const float (&in)[V]) {
  if constexpr (V == 1) {
    p[0] = from_f32<T>(in[0]);
  } else {
    Pack<T, V> raw;
#pragma unroll
    for (int i = 0; i < V; ++i) raw.v[i] = from_f32<T>(in[i]);
    *reinterpret_cast<Pack<T, V>*>(p) = raw;
  }
}

// Sum / max over the G lanes that share a row (G power of two, <= 64).  After the
// butterfly every lane of the group holds the result.
__device__ __forceinline__ float group_sum(float v, int G) {
  for (int o = G >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float group_max(float v, int G) {
  for (int o = G >> 1; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float wave_sum(float v) { return group_sum(v, 64); }

// Block-level sum of N per-thread accumulators -> partials[blockIdx.x * N + k].
// Fixed order (lane butterfly, then waves 0..3), so results are reproducible.
template <int N>
__device__ __forceinline__ void block_partials(float (&acc)[N], float* __restrict__ partials) {
  __shared__ float lds[kBlock / 64][N];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < N; ++k) {
    float s = wave_sum(acc[k]);
    if (lane == 0) lds[wave][k] = s;
  }
  __syncthreads();
  if (threadIdx.x < N) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < kBlock / 64; ++w) s += lds[w][threadIdx.x];
    partials[(size_t)blockIdx.x * N + threadIdx.x] = s;
  }
}

// Philox4x32-10 counter RNG (Salmon et al. 2011), used when the caller passes no
// noise tensor.  Counter = global element index, key = seed, so a sample's noise
// does not depend on how the batch is sharded over ranks.
struct Philox {
  static constexpr uint32_t kM0 = 0xD2511F53u, kM1 = 0xCD9E8D57u;
  static constexpr uint32_t kW0 = 0x9E3779B9u, kW1 = 0xBB67AE85u;
  __device__ __forceinline__ static void gen(uint64_t seed, uint64_t ctr_lo, uint64_t ctr_hi, uint32_t (&out)[4]) {
    uint32_t c0 = (uint32_t)ctr_lo, c1 = (uint32_t)(ctr_lo >> 32);
    uint32_t c2 = (uint32_t)ctr_hi, c3 = (uint32_t)(ctr_hi >> 32);
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
      uint32_t hi0 = __umulhi(kM0, c0), lo0 = kM0 * c0;
      uint32_t hi1 = __umulhi(kM1, c2), lo1 = kM1 * c2;
      uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
      c0 = n0; c1 = n1; c2 = n2; c3 = n3;
      k0 += kW0; k1 += kW1;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
  }
};

// uniform in (0,1): 24 random bits, centred so 0 and 1 are never produced
__device__ __forceinline__ float u01(uint32_t x) { return ((float)(x >> 8) + 0.5f) * (1.0f / 16777216.0f); }

// four N(0,1) draws from four words (two Box-Muller pairs)
__device__ __forceinline__ void normal4(const uint32_t (&w)[4], float (&n)[4]) {
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    float r = sqrtf(-2.0f * __logf(u01(w[2 * p])));
    float s, c;
    __sincosf(6.283185307179586f * u01(w[2 * p + 1]), &s, &c);
    n[2 * p] = r * c;
    n[2 * p + 1] = r * s;
  }
}

// V draws for the elements [elem, elem+V) of a [rows, cols] noise field (elem is
// the global flat index of the first element; V is 1, 4 or 8 and elem % V == 0
// for V > 1).  The stream id separates independent fields of one call.
template <int V>
__device__ __forceinline__ void philox_normal(uint64_t seed, uint64_t stream, uint64_t elem, float (&out)[V]) {
  if constexpr (V == 1) {
    uint32_t w[4];
    float n[4];
    Philox::gen(seed, elem >> 2, stream, w);
    normal4(w, n);
    out[0] = n[elem & 3];
  } else {
#pragma unroll
    for (int q = 0; q < V / 4; ++q) {
      uint32_t w[4];
      float n[4];
      Philox::gen(seed, (elem >> 2) + q, stream, w);
      normal4(w, n);
#pragma unroll
      for (int i = 0; i < 4; ++i) out[4 * q + i] = n[i];
    }
  }
}
template <int V>
__device__ __forceinline__ void philox_uniform(uint64_t seed, uint64_t stream, uint64_t elem, float (&out)[V]) {
  if constexpr (V == 1) {
    uint32_t w[4];
    Philox::gen(seed, elem >> 2, stream, w);
    out[0] = u01(w[elem & 3]);
  } else {
#pragma unroll
    for (int q = 0; q < V / 4; ++q) {
      uint32_t w[4];
      Philox::gen(seed, (elem >> 2) + q, stream, w);
#pragma unroll
      for (int i = 0; i < 4; ++i) out[4 * q + i] = u01(w[i]);
    }
  }
}

// ---- host-side launch geometry -------------------------------------------------
struct RowGeom {
  int G;               // lanes per row
  int rows_per_block;  // kBlock / G
  int64_t tiles;       // ceil(rows / rows_per_block)
  int grid;            // min(tiles, kMaxGrid)
};

inline RowGeom row_geom(int64_t rows, int64_t vecs_per_row) {
  RowGeom g;
  int G = 1;
  while (G < 64 && G < vecs_per_row) G <<= 1;
  g.G = G;
  g.rows_per_block = kBlock / G;
  g.tiles = (rows + g.rows_per_block - 1) / g.rows_per_block;
  g.grid = (int)(g.tiles < kMaxGrid ? g.tiles : kMaxGrid);
  if (g.grid < 1) g.grid = 1;
  return g;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace vae
