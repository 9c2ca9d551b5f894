// Beta reparameterizer: concentration split + KL (forward and backward).  The sample itself stays
// torch._sample_dirichlet (rejection sampler, not injectable).
//
// Replaces the op chains of /root/reference reparameterizers/beta.py:
//   :105-106  conc1 = sigmoid(logits[..., :L]); conc2 = sigmoid(logits[..., L:])
//   :110-136  KL(Beta(conc1, conc2) || Beta(1.1, 1.1) or a prior) summed over -1, i.e.
//             torch/distributions/kl.py:219-228 (_kl_beta_beta): with sp = a1 + a0, sq = b1 + b0
//               lgamma(b1) + lgamma(b0) + lgamma(sp) - lgamma(a1) - lgamma(a0) - lgamma(sq)
//               + (a1 - b1) psi(a1) + (a0 - b0) psi(a0) + (sq - sp) psi(sp)
//   :175-176  conc1_mean, conc2_mean
// 6 lgamma + 3 digamma per element: still one pass over HBM (8L in, 8L + 4 out per sample).
#include "common.hpp"
#include "../../include/vae_amd.h"

namespace vae {

constexpr float kBetaPrior = 1.1f;

// digamma / trigamma for x > 0: recurrence up to x >= 6, then the asymptotic series.
__device__ __forceinline__ float digammaf_(float x) {
  float r = 0.f;
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    if (x < 6.0f) {
      r -= 1.0f / x;
      x += 1.0f;
    }
  }
  const float i = 1.0f / x, i2 = i * i;
  return r + logf(x) - 0.5f * i - i2 * (1.0f / 12.0f - i2 * (1.0f / 120.0f - i2 * (1.0f / 252.0f)));
}
__device__ __forceinline__ float trigammaf_(float x) {
  float r = 0.f;
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    if (x < 6.0f) {
      r += 1.0f / (x * x);
      x += 1.0f;
    }
  }
  const float i = 1.0f / x, i2 = i * i;
  return r + i + 0.5f * i2 + i * i2 * (1.0f / 6.0f - i2 * (1.0f / 30.0f - i2 * (1.0f / 42.0f)));
}

__device__ __forceinline__ float beta_kl_elem(float a1, float a0, float b1, float b0) {
  const float sp = a1 + a0, sq = b1 + b0;
  const float t1 = lgammaf(b1) + lgammaf(b0) + lgammaf(sp);
  const float t2 = lgammaf(a1) + lgammaf(a0) + lgammaf(sq);
  return t1 - t2 + (a1 - b1) * digammaf_(a1) + (a0 - b0) * digammaf_(a0) + (sq - sp) * digammaf_(sp);
}
// d kl / d(a1, a0) and (optionally) d kl / d(b1, b0)
__device__ __forceinline__ void beta_kl_grad(float a1, float a0, float b1, float b0, float& ga1, float& ga0,
                                             float& gb1, float& gb0, bool prior_grads) {
  const float sp = a1 + a0, sq = b1 + b0;
  const float tsp = (sq - sp) * trigammaf_(sp);
  ga1 = (a1 - b1) * trigammaf_(a1) + tsp;
  ga0 = (a0 - b0) * trigammaf_(a0) + tsp;
  if (prior_grads) {
    const float psp = digammaf_(sp), psq = digammaf_(sq);
    gb1 = digammaf_(b1) - psq - digammaf_(a1) + psp;
    gb0 = digammaf_(b0) - psq - digammaf_(a0) + psp;
  }
}
__device__ __forceinline__ float sigmoid_b(float v) { return 1.0f / (1.0f + expf(-v)); }

// ---- fused: logits [B, 2L] -> conc1, conc0 [B, L], kl [B] vs Beta(1.1, 1.1), partial sums of conc
template <typename T, int V, int U>
__global__ __launch_bounds__(kBlock) void beta_split_fwd_kernel(
    const T* __restrict__ logits, int64_t row_stride, T* __restrict__ c1, T* __restrict__ c0, T* __restrict__ kl,
    float* __restrict__ ws, float mean_scale, float* __restrict__ means, int64_t B, int L, int G, int64_t tiles) {
  const int rpb = kBlock / G;
  const int sub = threadIdx.x % G;
  const int rloc = threadIdx.x / G;
  const int nvec = L / V;
  float acc[2] = {0.f, 0.f};
  for (int64_t t = blockIdx.x; t < tiles; t += gridDim.x) {
    const int64_t row = t * rpb + rloc;
    float klr = 0.f;
    if (row < B) {
      const T* lrow = logits + row * row_stride;
      for (int j0 = sub; j0 < nvec; j0 += G * U) {
        float a[U][V], b[U][V];
        load_chunk<T, V, U>(lrow, j0, G, nvec, a);
        load_chunk<T, V, U>(lrow + L, j0, G, nvec, b);
#pragma unroll
        for (int k = 0; k < U; ++k) {
          const int j = j0 + k * G;
          if (j < nvec) {
#pragma unroll
            for (int i = 0; i < V; ++i) {
              a[k][i] = to_f32(from_f32<T>(sigmoid_b(a[k][i])));
              b[k][i] = to_f32(from_f32<T>(sigmoid_b(b[k][i])));
              klr += beta_kl_elem(a[k][i], b[k][i], kBetaPrior, kBetaPrior);
              acc[0] += a[k][i];
              acc[1] += b[k][i];
            }
            store_vec<T, V>(c1 + row * L + (int64_t)j * V, a[k]);
            store_vec<T, V>(c0 + row * L + (int64_t)j * V, b[k]);
          }
        }
      }
    }
    klr = group_sum(klr, G);
    if (row < B && sub == 0) kl[row] = from_f32<T>(klr);
  }
  if (ws) grid_reduce<2>(acc, ws, mean_scale, means);
}

// glogits = sigmoid' * (g_conc + gkl * dKL/dconc + g_mean * mean_scale)
template <typename T, int V, int U>
__global__ __launch_bounds__(kBlock) void beta_split_bwd_kernel(
    const T* __restrict__ logits, int64_t row_stride, const T* __restrict__ gc1, const T* __restrict__ gc0,
    const T* __restrict__ gkl, const float* __restrict__ gmeans, T* __restrict__ glogits, int64_t B, int L,
    float mean_scale, int G, int64_t tiles) {
  const int rpb = kBlock / G;
  const int sub = threadIdx.x % G;
  const int rloc = threadIdx.x / G;
  const int nvec = L / V;
  const float gm1 = gmeans ? gmeans[0] * mean_scale : 0.f;
  const float gm0 = gmeans ? gmeans[1] * mean_scale : 0.f;
  for (int64_t t = blockIdx.x; t < tiles; t += gridDim.x) {
    const int64_t row = t * rpb + rloc;
    if (row >= B) continue;
    const T* lrow = logits + row * row_stride;
    T* grow = glogits + row * 2 * (int64_t)L;
    const float gk = gkl ? to_f32(gkl[row]) : 0.f;
    for (int j0 = sub; j0 < nvec; j0 += G * U) {
      float a[U][V], b[U][V], g1[U][V], g0[U][V];
      load_chunk<T, V, U>(lrow, j0, G, nvec, a);
      load_chunk<T, V, U>(lrow + L, j0, G, nvec, b);
      if (gc1) load_chunk<T, V, U>(gc1 + row * L, j0, G, nvec, g1);
      if (gc0) load_chunk<T, V, U>(gc0 + row * L, j0, G, nvec, g0);
#pragma unroll
      for (int k = 0; k < U; ++k) {
        const int j = j0 + k * G;
        if (j < nvec) {
          float o1[V], o0[V];
#pragma unroll
          for (int i = 0; i < V; ++i) {
            const float s1 = to_f32(from_f32<T>(sigmoid_b(a[k][i])));
            const float s0 = to_f32(from_f32<T>(sigmoid_b(b[k][i])));
            float d1 = 0.f, d0 = 0.f, u1, u0;
            if (gkl) beta_kl_grad(s1, s0, kBetaPrior, kBetaPrior, d1, d0, u1, u0, false);
            o1[i] = s1 * (1.0f - s1) * ((gc1 ? g1[k][i] : 0.f) + gk * d1 + gm1);
            o0[i] = s0 * (1.0f - s0) * ((gc0 ? g0[k][i] : 0.f) + gk * d0 + gm0);
          }
          store_vec<T, V>(grow + (int64_t)j * V, o1);
          store_vec<T, V>(grow + L + (int64_t)j * V, o0);
        }
      }
    }
  }
}

// ---- standalone KL between two Betas given as strided [B, L] concentration views
template <typename T, int V, int U, bool BWD>
__global__ __launch_bounds__(kBlock) void beta_kl_kernel(
    const T* __restrict__ a1, int64_t sa1, const T* __restrict__ a0, int64_t sa0, const T* __restrict__ b1,
    int64_t sb1, const T* __restrict__ b0, int64_t sb0, const T* __restrict__ gkl, T* __restrict__ kl,
    T* __restrict__ ga1, T* __restrict__ ga0, T* __restrict__ gb1, T* __restrict__ gb0, int64_t B, int L, int G,
    int64_t tiles) {
  const int rpb = kBlock / G;
  const int sub = threadIdx.x % G;
  const int rloc = threadIdx.x / G;
  const int nvec = L / V;
  for (int64_t t = blockIdx.x; t < tiles; t += gridDim.x) {
    const int64_t row = t * rpb + rloc;
    float acc = 0.f;
    if (row < B) {
      const float gk = BWD ? to_f32(gkl[row]) : 0.f;
      for (int j0 = sub; j0 < nvec; j0 += G * U) {
        float x1[U][V], x0[U][V], y1[U][V], y0[U][V];
        load_chunk<T, V, U>(a1 + row * sa1, j0, G, nvec, x1);
        load_chunk<T, V, U>(a0 + row * sa0, j0, G, nvec, x0);
        if (b1) {
          load_chunk<T, V, U>(b1 + row * sb1, j0, G, nvec, y1);
          load_chunk<T, V, U>(b0 + row * sb0, j0, G, nvec, y0);
        }
#pragma unroll
        for (int k = 0; k < U; ++k) {
          const int j = j0 + k * G;
          if (j < nvec) {
            float o1[V], o0[V], p1[V], p0[V];
#pragma unroll
            for (int i = 0; i < V; ++i) {
              const float q1 = b1 ? y1[k][i] : kBetaPrior, q0 = b1 ? y0[k][i] : kBetaPrior;
              if constexpr (!BWD) {
                acc += beta_kl_elem(x1[k][i], x0[k][i], q1, q0);
              } else {
                beta_kl_grad(x1[k][i], x0[k][i], q1, q0, o1[i], o0[i], p1[i], p0[i], gb1 != nullptr);
                o1[i] *= gk; o0[i] *= gk;
                if (gb1) { p1[i] *= gk; p0[i] *= gk; }
              }
            }
            if constexpr (BWD) {
              const int64_t o = row * L + (int64_t)j * V;
              store_vec<T, V>(ga1 + o, o1);
              store_vec<T, V>(ga0 + o, o0);
              if (gb1) {
                store_vec<T, V>(gb1 + o, p1);
                store_vec<T, V>(gb0 + o, p0);
              }
            }
          }
        }
      }
    }
    if constexpr (!BWD) {
      acc = group_sum(acc, G);
      if (row < B && sub == 0) kl[row] = from_f32<T>(acc);
    }
  }
}


template <typename T>
static int beta_split_fwd_dispatch(const void* logits, int64_t rs, void* c1, void* c0, void* kl, float* means,
                                   float* workspace, int64_t B, int64_t L, float mean_scale, hipStream_t stream) {
  constexpr int VN = VecOf<T>::n;
  const bool vec = (L % VN == 0) && (rs % VN == 0) && aligned16(logits) && aligned16(c1) && aligned16(c0);
  const RowGeom g = row_geom(B, vec ? L / VN : L, kGridReduce);
  if (vec) {
    VAE_SWITCH_U(g.U, hipLaunchKernelGGL((beta_split_fwd_kernel<T, VN, U>), dim3(g.grid), dim3(kBlock), 0, stream,
                                         (const T*)logits, rs, (T*)c1, (T*)c0, (T*)kl, means ? workspace : nullptr,
                                         mean_scale, means, B, (int)L, g.G, g.tiles));
  } else {
    VAE_SWITCH_U(g.U, hipLaunchKernelGGL((beta_split_fwd_kernel<T, 1, U>), dim3(g.grid), dim3(kBlock), 0, stream,
                                         (const T*)logits, rs, (T*)c1, (T*)c0, (T*)kl, means ? workspace : nullptr,
                                         mean_scale, means, B, (int)L, g.G, g.tiles));
  }
  return hipGetLastError() == hipSuccess ? VAE_OK : VAE_ERR_LAUNCH;
}

template <typename T>
static int beta_split_bwd_dispatch(const void* logits, int64_t rs, const void* gc1, const void* gc0, const void* gkl,
                                   const float* gmeans, void* glogits, int64_t B, int64_t L, float mean_scale,
                                   hipStream_t stream) {
  constexpr int VN = VecOf<T>::n;
  const bool vec = (L % VN == 0) && (rs % VN == 0) && aligned16(logits) && aligned16(glogits) &&
                   (!gc1 || aligned16(gc1)) && (!gc0 || aligned16(gc0));
  const RowGeom g = row_geom(B, vec ? L / VN : L, kGridStream);
  if (vec) {
    VAE_SWITCH_U(g.U, hipLaunchKernelGGL((beta_split_bwd_kernel<T, VN, U>), dim3(g.grid), dim3(kBlock), 0, stream,
                                         (const T*)logits, rs, (const T*)gc1, (const T*)gc0, (const T*)gkl, gmeans,
                                         (T*)glogits, B, (int)L, mean_scale, g.G, g.tiles));
  } else {
    VAE_SWITCH_U(g.U, hipLaunchKernelGGL((beta_split_bwd_kernel<T, 1, U>), dim3(g.grid), dim3(kBlock), 0, stream,
                                         (const T*)logits, rs, (const T*)gc1, (const T*)gc0, (const T*)gkl, gmeans,
                                         (T*)glogits, B, (int)L, mean_scale, g.G, g.tiles));
  }
  return hipGetLastError() == hipSuccess ? VAE_OK : VAE_ERR_LAUNCH;
}

template <typename T, bool BWD>
static int beta_kl_dispatch(const void* a1, int64_t sa1, const void* a0, int64_t sa0, const void* b1, int64_t sb1,
                            const void* b0, int64_t sb0, const void* gkl, void* kl, void* ga1, void* ga0, void* gb1,
                            void* gb0, int64_t B, int64_t L, hipStream_t stream) {
  constexpr int VN = VecOf<T>::n;
  bool vec = (L % VN == 0) && (sa1 % VN == 0) && (sa0 % VN == 0) && aligned16(a1) && aligned16(a0);
  if (b1) vec = vec && (sb1 % VN == 0) && (sb0 % VN == 0) && aligned16(b1) && aligned16(b0);
  if (BWD) vec = vec && aligned16(ga1) && aligned16(ga0) && (!gb1 || (aligned16(gb1) && aligned16(gb0)));
  const RowGeom g = row_geom(B, vec ? L / VN : L, kGridStream);
  if (vec) {
    VAE_SWITCH_U(g.U, hipLaunchKernelGGL((beta_kl_kernel<T, VN, U, BWD>), dim3(g.grid), dim3(kBlock), 0, stream,
                                         (const T*)a1, sa1, (const T*)a0, sa0, (const T*)b1, sb1, (const T*)b0, sb0,
                                         (const T*)gkl, (T*)kl, (T*)ga1, (T*)ga0, (T*)gb1, (T*)gb0, B, (int)L, g.G,
                                         g.tiles));
  } else {
    VAE_SWITCH_U(g.U, hipLaunchKernelGGL((beta_kl_kernel<T, 1, U, BWD>), dim3(g.grid), dim3(kBlock), 0, stream,
                                         (const T*)a1, sa1, (const T*)a0, sa0, (const T*)b1, sb1, (const T*)b0, sb0,
                                         (const T*)gkl, (T*)kl, (T*)ga1, (T*)ga0, (T*)gb1, (T*)gb0, B, (int)L, g.G,
                                         g.tiles));
  }
  return hipGetLastError() == hipSuccess ? VAE_OK : VAE_ERR_LAUNCH;
}

}  // namespace vae

using namespace vae;

extern "C" {

int vae_beta_split_kl_fwd(const void* logits, int64_t row_stride, void* conc1, void* conc2, void* kl, float* means,
                          float* workspace, int64_t B, int64_t L, float mean_scale, int dtype, vae_stream_t stream) {
  if (B < 0 || L <= 0 || L > VAE_MAX_ROW) return VAE_ERR_ARG;
  if (B == 0) return VAE_OK;
  if (!logits || !conc1 || !conc2 || !kl || !workspace || row_stride < 2 * L) return VAE_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == VAE_F32)
    return beta_split_fwd_dispatch<float>(logits, row_stride, conc1, conc2, kl, means, workspace, B, L, mean_scale, st);
  if (dtype == VAE_BF16)
    return beta_split_fwd_dispatch<bf16_t>(logits, row_stride, conc1, conc2, kl, means, workspace, B, L, mean_scale, st);
  return VAE_ERR_ARG;
}

int vae_beta_split_kl_bwd(const void* logits, int64_t row_stride, const void* gconc1, const void* gconc2,
                          const void* gkl, const float* gmeans, void* glogits, int64_t B, int64_t L, float mean_scale,
                          int dtype, vae_stream_t stream) {
  if (B < 0 || L <= 0 || L > VAE_MAX_ROW) return VAE_ERR_ARG;
  if (B == 0) return VAE_OK;
  if (!logits || !glogits || row_stride < 2 * L) return VAE_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == VAE_F32)
    return beta_split_bwd_dispatch<float>(logits, row_stride, gconc1, gconc2, gkl, gmeans, glogits, B, L, mean_scale, st);
  if (dtype == VAE_BF16)
    return beta_split_bwd_dispatch<bf16_t>(logits, row_stride, gconc1, gconc2, gkl, gmeans, glogits, B, L, mean_scale, st);
  return VAE_ERR_ARG;
}

int vae_beta_kl_fwd(const void* a1, int64_t stride_a1, const void* a0, int64_t stride_a0, const void* b1,
                    int64_t stride_b1, const void* b0, int64_t stride_b0, void* kl, int64_t B, int64_t L, int dtype,
                    vae_stream_t stream) {
  if (B < 0 || L <= 0 || L > VAE_MAX_ROW) return VAE_ERR_ARG;
  if (B == 0) return VAE_OK;
  if (!a1 || !a0 || !kl || ((b1 == nullptr) != (b0 == nullptr))) return VAE_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == VAE_F32)
    return beta_kl_dispatch<float, false>(a1, stride_a1, a0, stride_a0, b1, stride_b1, b0, stride_b0, nullptr, kl,
                                          nullptr, nullptr, nullptr, nullptr, B, L, st);
  if (dtype == VAE_BF16)
    return beta_kl_dispatch<bf16_t, false>(a1, stride_a1, a0, stride_a0, b1, stride_b1, b0, stride_b0, nullptr, kl,
                                           nullptr, nullptr, nullptr, nullptr, B, L, st);
  return VAE_ERR_ARG;
}

int vae_beta_kl_bwd(const void* a1, int64_t stride_a1, const void* a0, int64_t stride_a0, const void* b1,
                    int64_t stride_b1, const void* b0, int64_t stride_b0, const void* gkl, void* ga1, void* ga0,
                    void* gb1, void* gb0, int64_t B, int64_t L, int dtype, vae_stream_t stream) {
  if (B < 0 || L <= 0 || L > VAE_MAX_ROW) return VAE_ERR_ARG;
  if (B == 0) return VAE_OK;
  if (!a1 || !a0 || !gkl || !ga1 || !ga0 || ((b1 == nullptr) != (b0 == nullptr)) ||
      ((gb1 == nullptr) != (gb0 == nullptr)) || (gb1 && !b1))
    return VAE_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == VAE_F32)
    return beta_kl_dispatch<float, true>(a1, stride_a1, a0, stride_a0, b1, stride_b1, b0, stride_b0, gkl, nullptr, ga1,
                                         ga0, gb1, gb0, B, L, st);
  if (dtype == VAE_BF16)
    return beta_kl_dispatch<bf16_t, true>(a1, stride_a1, a0, stride_a0, b1, stride_b1, b0, stride_b0, gkl, nullptr,
                                          ga1, ga0, gb1, gb0, B, L, st);
  return VAE_ERR_ARG;
}

}  // extern "C"
