// Relaxed-Bernoulli sample (soft + straight-through hard) and the eps-stabilised Bernoulli KL.
//
// Replaces the op chains of /root/reference reparameterizers/bernoulli.py:
//   :134      D.RelaxedBernoulli(tau, logits=l).rsample(), i.e. (torch/distributions/
//             relaxed_bernoulli.py:104-112, utils.py:101-124)
//               p = clamp(sigmoid(l), e, 1-e); u = clamp(U, e, 1-e), e = finfo.eps
//               y = sigmoid((log u - log1p(-u) + log p - log1p(-p)) / tau)
//   :111-124  hard = (y >= 0.5); z_hard = (hard - y).detach() + y
//   :147-191  KL = sum q1*(log(q1+1e-6) - log(p1+1e-6)) + q0*(log(q0+1e-6) - log(p0+1e-6)),
//             p = 0.5 when no prior logits are given
// The sample is elementwise (rows only matter for the Philox counter); the KL is a row sum.
#include "common.hpp"
#include "../../include/vae_amd.h"

namespace vae {

template <typename T>
struct ClampEps;
template <>
struct ClampEps<float> {
  static constexpr float v = 1.1920928955078125e-07f;  // torch.finfo(float32).eps
};
template <>
struct ClampEps<bf16_t> {
  static constexpr float v = 0.0078125f;  // torch.finfo(bfloat16).eps
};
template <>
struct ClampEps<f16_t> {
  static constexpr float v = 0.0009765625f;  // torch.finfo(float16).eps
};
constexpr float kBernKlEps = 1e-6f;

__device__ __forceinline__ float sigmoid_div(float v) { return rcp_nr(1.0f + exp_acc(-v)); }

template <typename T, int V, int U>
__global__ __launch_bounds__(kBlock) void bernoulli_fwd_kernel(
    const T* __restrict__ logits, int64_t row_stride, const T* __restrict__ unif, T* __restrict__ y_out,
    int64_t y_stride, T* __restrict__ zh_out, int64_t zh_stride, int64_t B, int K, float tau, uint64_t seed, uint64_t offset, int64_t row0, int G,
    int64_t tiles) {
  offset = resolve_offset(offset);
  const int rpb = kBlock / G;
  const bool nts = nt_stores((int64_t)B * K * (int64_t)sizeof(T));
  const int sub = threadIdx.x % G;
  const int rloc = threadIdx.x / G;
  const int nvec = K / V;
  constexpr float e = ClampEps<T>::v;
  const float itau = 1.0f / tau;
  for (int64_t t = blockIdx.x; t < tiles; t += gridDim.x) {
    const int64_t row = t * rpb + rloc;
    if (row >= B) continue;
    for (int j0 = sub; j0 < nvec; j0 += G * U) {
      float l[U][V], u[U][V];
      load_chunk<T, V, U>(logits + row * row_stride, j0, G, nvec, l);
      if (unif) load_chunk<T, V, U>(unif + row * K, j0, G, nvec, u);
#pragma unroll
      for (int k = 0; k < U; ++k) {
        const int j = j0 + k * G;
        if (j < nvec) {
          if (!unif) philox_uniform<V>(seed, offset, (uint64_t)(row0 + row) * (uint64_t)K + (uint64_t)j * V, u[k]);
          float y[V], zh[V];
#pragma unroll
          for (int i = 0; i < V; ++i) {
            const float p = fminf(fmaxf(to_f32(from_f32<T>(sigmoid_div(l[k][i]))), e), 1.0f - e);
            const float uu = fminf(fmaxf(u[k][i], e), 1.0f - e);
            // all four arguments lie in [eps, 1 - eps] (normal range): bare v_log_f32, and the two differences as
            // differences of log2 so that one multiply by ln 2 / tau serves the whole logit
            const float lg = ((log2_raw(uu) - log2_raw(1.0f - uu)) + (log2_raw(p) - log2_raw(1.0f - p))) * (kLn2 * itau);
            y[i] = to_f32(from_f32<T>(sigmoid_div(lg)));
            const float h = y[i] >= 0.5f ? 1.0f : 0.0f;
            zh[i] = add_rn(to_f32(from_f32<T>(sub_rn(h, y[i]))), y[i]);
          }
          store_vec<T, V>(nts, y_out + row * y_stride + (int64_t)j * V, y);
          if (zh_out) store_vec<T, V>(nts, zh_out + row * zh_stride + (int64_t)j * V, zh);
        }
      }
    }
  }
}

// d y / d l = y (1 - y) / tau wherever sigmoid(l) was not clamped (the clamp passes no gradient)
template <typename T, int V, int U>
__global__ __launch_bounds__(kBlock) void bernoulli_bwd_kernel(
    const T* __restrict__ logits, int64_t row_stride, const T* __restrict__ y, int64_t y_stride,
    const T* __restrict__ gy, int64_t gy_stride, const T* __restrict__ gzh, int64_t gzh_stride,
    T* __restrict__ glogits, int64_t gl_stride, int64_t B, int K, float tau, int G, int64_t tiles) {
  const int rpb = kBlock / G;
  const bool nts = nt_stores((int64_t)B * K * (int64_t)sizeof(T));
  const int sub = threadIdx.x % G;
  const int rloc = threadIdx.x / G;
  const int nvec = K / V;
  constexpr float e = ClampEps<T>::v;
  const float itau = 1.0f / tau;
  for (int64_t t = blockIdx.x; t < tiles; t += gridDim.x) {
    const int64_t row = t * rpb + rloc;
    if (row >= B) continue;
    for (int j0 = sub; j0 < nvec; j0 += G * U) {
      float l[U][V], yv[U][V], g1[U][V], g2[U][V];
      load_chunk<T, V, U>(logits + row * row_stride, j0, G, nvec, l);
      load_chunk<T, V, U>(y + row * y_stride, j0, G, nvec, yv);
      if (gy) load_chunk<T, V, U>(gy + row * gy_stride, j0, G, nvec, g1);
      if (gzh) load_chunk<T, V, U>(gzh + row * gzh_stride, j0, G, nvec, g2);
#pragma unroll
      for (int k = 0; k < U; ++k) {
        const int j = j0 + k * G;
        if (j < nvec) {
          float o[V];
#pragma unroll
          for (int i = 0; i < V; ++i) {
            const float s = to_f32(from_f32<T>(sigmoid_div(l[k][i])));
            const float pass = (s >= e && s <= 1.0f - e) ? 1.0f : 0.0f;
            const float g = (gy ? g1[k][i] : 0.f) + (gzh ? g2[k][i] : 0.f);
            o[i] = pass * g * yv[k][i] * (1.0f - yv[k][i]) * itau;
          }
          store_vec<T, V>(nts, glogits + row * gl_stride + (int64_t)j * V, o);
        }
      }
    }
  }
}

// q via t = exp(-|l|), r = 1/(1+t): the larger of (q1, q0) is r, the smaller t*r (no 1-q cancellation, one
// exp + one rcp for both); with no prior the summand is symmetric in the sign of l, so no select is needed.
// Two logs per element remain because of the +1e-6 inside them.  HASP = prior logits given.
template <typename T, int V, int U, bool BWD, bool HASP>
__global__ __launch_bounds__(kBlock) void bernoulli_kl_kernel(
    const T* __restrict__ lq, int64_t stride_q, const T* __restrict__ lp, int64_t stride_p,
    const T* __restrict__ gkl, T* __restrict__ kl, T* __restrict__ gq, T* __restrict__ gp, int64_t B, int K, int G,
    int64_t tiles) {
  const int rpb = kBlock / G;
  const bool nts = nt_stores((int64_t)B * K * (int64_t)sizeof(T));
  const int sub = threadIdx.x % G;
  const int rloc = threadIdx.x / G;
  const int nvec = K / V;
  const float lhalf = __logf(0.5f + kBernKlEps);  // both prior terms when no prior logits are given
  for (int64_t t = blockIdx.x; t < tiles; t += gridDim.x) {
    const int64_t row = t * rpb + rloc;
    float acc = 0.f;
    if (row < B) {
      const float gk = BWD ? to_f32(gkl[row]) : 0.f;
      for (int j0 = sub; j0 < nvec; j0 += G * U) {
        float a[U][V], b[U][V];
        load_chunk<T, V, U>(lq + row * stride_q, j0, G, nvec, a);
        if constexpr (HASP) load_chunk<T, V, U>(lp + row * stride_p, j0, G, nvec, b);
#pragma unroll
        for (int k = 0; k < U; ++k) {
          const int j = j0 + k * G;
          if (j < nvec) {
            float oq[V], op[V];
#pragma unroll
            for (int i = 0; i < V; ++i) {
              const float tq = __expf(-fabsf(a[k][i])), rq = fast_rcp(1.0f + tq);
              const float qb = rq, qs = tq * rq;  // larger / smaller of (q1, q0)
              const float lb = fast_log(qb + kBernKlEps), ls = fast_log(qs + kBernKlEps);
              if constexpr (!HASP) {
                if constexpr (!BWD) {
                  acc += qb * (lb - lhalf) + qs * (ls - lhalf);
                } else {
                  // d/dl of the summand, written for l >= 0 (q1 = qb) and mirrored for l < 0
                  const float d = (lb + qb * fast_rcp(qb + kBernKlEps)) - (ls + qs * fast_rcp(qs + kBernKlEps));
                  oq[i] = gk * copysignf(d * qb * qs, a[k][i]);
                }
              } else {
                const bool posq = a[k][i] >= 0.f;
                const float q1 = posq ? qb : qs, q0 = posq ? qs : qb;
                const float lq1 = posq ? lb : ls, lq0 = posq ? ls : lb;
                const float tp = __expf(-fabsf(b[k][i])), rp = fast_rcp(1.0f + tp);
                const bool posp = b[k][i] >= 0.f;
                const float p1 = posp ? rp : tp * rp, p0 = posp ? tp * rp : rp;
                const float d1 = lq1 - fast_log(p1 + kBernKlEps), d0 = lq0 - fast_log(p0 + kBernKlEps);
                if constexpr (!BWD) {
                  acc += q1 * d1 + q0 * d0;
                } else {
                  const float dq1 = d1 + q1 * fast_rcp(q1 + kBernKlEps) - d0 - q0 * fast_rcp(q0 + kBernKlEps);
                  oq[i] = gk * dq1 * q1 * q0;
                  op[i] = gk * (q0 * fast_rcp(p0 + kBernKlEps) - q1 * fast_rcp(p1 + kBernKlEps)) * p1 * p0;
                }
              }
            }
            if constexpr (BWD) {
              store_vec<T, V>(nts, gq + row * K + (int64_t)j * V, oq);
              if constexpr (HASP) {
                if (gp) store_vec<T, V>(nts, gp + row * K + (int64_t)j * V, op);
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

#define BERN_LAUNCH(KERNEL, vec_, g_, ...)                                                           \
  if (vec_) {                                                                                        \
    VAE_SWITCH_U(g_.U, hipLaunchKernelGGL((KERNEL<T, VN, U>), dim3(g_.grid), dim3(kBlock), 0, stream, __VA_ARGS__)); \
  } else {                                                                                           \
    VAE_SWITCH_U(g_.U, hipLaunchKernelGGL((KERNEL<T, 1, U>), dim3(g_.grid), dim3(kBlock), 0, stream, __VA_ARGS__));  \
  }

template <typename T>
static int bern_fwd_dispatch(const void* logits, int64_t rs, const void* unif, void* y, int64_t ys, void* zh,
                             int64_t zs, int64_t B,
                             int64_t K, float tau, uint64_t seed, uint64_t offset, int64_t row0, hipStream_t stream) {
  constexpr int VN = VecOf<T>::n;
  const bool vec = (K % VN == 0) && (rs % VN == 0) && (ys % VN == 0) && (zs % VN == 0) && aligned16(logits) &&
                   aligned16(y) && aligned16(zh) && (!unif || aligned16(unif));
  const RowGeom g = row_geom(B, vec ? K / VN : K, kGridStream);
  BERN_LAUNCH(bernoulli_fwd_kernel, vec, g, (const T*)logits, rs, (const T*)unif, (T*)y, ys, (T*)zh, zs, B, (int)K, tau, seed,
              offset, row0, g.G, g.tiles);
  return hipGetLastError() == hipSuccess ? VAE_OK : VAE_ERR_LAUNCH;
}

template <typename T>
static int bern_bwd_dispatch(const void* logits, int64_t rs, const void* y, int64_t ys, const void* gy, int64_t gys,
                             const void* gzh, int64_t gzs, void* glogits, int64_t gls, int64_t B, int64_t K, float tau, hipStream_t stream) {
  constexpr int VN = VecOf<T>::n;
  const bool vec = (K % VN == 0) && (rs % VN == 0) && (ys % VN == 0) && (gys % VN == 0) && (gzs % VN == 0) &&
                   (gls % VN == 0) && aligned16(logits) && aligned16(y) && aligned16(glogits) &&
                   (!gy || aligned16(gy)) && (!gzh || aligned16(gzh));
  const RowGeom g = row_geom(B, vec ? K / VN : K, kGridStream);
  BERN_LAUNCH(bernoulli_bwd_kernel, vec, g, (const T*)logits, rs, (const T*)y, ys, (const T*)gy, gys, (const T*)gzh, gzs,
              (T*)glogits, gls, B, (int)K, tau, g.G, g.tiles);
  return hipGetLastError() == hipSuccess ? VAE_OK : VAE_ERR_LAUNCH;
}

template <typename T, bool BWD>
static int bern_kl_dispatch(const void* lq, int64_t sq, const void* lp, int64_t sp, const void* gkl, void* kl,
                            void* gq, void* gp, int64_t B, int64_t K, hipStream_t stream) {
  constexpr int VN = VecOf<T>::n;
  bool vec = (K % VN == 0) && (sq % VN == 0) && aligned16(lq) && (!lp || ((sp % VN == 0) && aligned16(lp)));
  if (BWD) vec = vec && aligned16(gq) && (!gp || aligned16(gp));
  const RowGeom g = row_geom(B, vec ? K / VN : K, kGridStream);
#define BERN_KL_GO(VV, HP)                                                                                          \
  VAE_SWITCH_U(g.U, hipLaunchKernelGGL((bernoulli_kl_kernel<T, VV, U, BWD, HP>), dim3(g.grid), dim3(kBlock), 0, stream, \
                                       (const T*)lq, sq, (const T*)lp, sp, (const T*)gkl, (T*)kl, (T*)gq, (T*)gp, B,    \
                                       (int)K, g.G, g.tiles))
  if (vec) {
    if (lp) { BERN_KL_GO(VN, true); } else { BERN_KL_GO(VN, false); }
  } else {
    if (lp) { BERN_KL_GO(1, true); } else { BERN_KL_GO(1, false); }
  }
#undef BERN_KL_GO
  return hipGetLastError() == hipSuccess ? VAE_OK : VAE_ERR_LAUNCH;
}

}  // namespace vae

using namespace vae;

extern "C" {

int vae_bernoulli_fwd(const void* logits, int64_t row_stride, const void* unif, void* y, int64_t y_stride, void* z_hard,
                      int64_t z_hard_stride, int64_t B,
                      int64_t K, float tau, uint64_t seed, uint64_t offset, int64_t row0, int dtype,
                      vae_stream_t stream) {
  if (B < 0 || K <= 0 || K > VAE_MAX_ROW || !(tau > 0.f)) return VAE_ERR_ARG;
  if (B == 0) return VAE_OK;
  if (y_stride == 0) y_stride = K;
  if (z_hard_stride == 0) z_hard_stride = K;
  if (!logits || !y || row_stride < K || y_stride < K || z_hard_stride < K) return VAE_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == VAE_F32) return bern_fwd_dispatch<float>(logits, row_stride, unif, y, y_stride, z_hard, z_hard_stride, B, K, tau, seed, offset, row0, st);
  if (dtype == VAE_BF16) return bern_fwd_dispatch<bf16_t>(logits, row_stride, unif, y, y_stride, z_hard, z_hard_stride, B, K, tau, seed, offset, row0, st);
  if (dtype == VAE_F16) return bern_fwd_dispatch<f16_t>(logits, row_stride, unif, y, y_stride, z_hard, z_hard_stride, B, K, tau, seed, offset, row0, st);
  return VAE_ERR_ARG;
}

int vae_bernoulli_bwd(const void* logits, int64_t row_stride, const void* y, int64_t y_stride, const void* gy,
                      int64_t gy_stride, const void* gz_hard, int64_t gz_hard_stride, void* glogits,
                      int64_t glogits_stride, int64_t B, int64_t K, float tau, int dtype, vae_stream_t stream) {
  if (B < 0 || K <= 0 || K > VAE_MAX_ROW || !(tau > 0.f)) return VAE_ERR_ARG;
  if (B == 0) return VAE_OK;
  if (y_stride == 0) y_stride = K;
  if (gy_stride == 0) gy_stride = K;
  if (gz_hard_stride == 0) gz_hard_stride = K;
  if (glogits_stride == 0) glogits_stride = K;
  if (!logits || !y || !glogits || row_stride < K || y_stride < K || gy_stride < K || gz_hard_stride < K ||
      glogits_stride < K)
    return VAE_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == VAE_F32) return bern_bwd_dispatch<float>(logits, row_stride, y, y_stride, gy, gy_stride, gz_hard,
                                                         gz_hard_stride, glogits, glogits_stride, B, K, tau, st);
  if (dtype == VAE_BF16) return bern_bwd_dispatch<bf16_t>(logits, row_stride, y, y_stride, gy, gy_stride, gz_hard,
                                                           gz_hard_stride, glogits, glogits_stride, B, K, tau, st);
  if (dtype == VAE_F16) return bern_bwd_dispatch<f16_t>(logits, row_stride, y, y_stride, gy, gy_stride, gz_hard,
                                                           gz_hard_stride, glogits, glogits_stride, B, K, tau, st);
  return VAE_ERR_ARG;
}

int vae_bernoulli_kl_fwd(const void* logits_q, int64_t stride_q, const void* logits_p, int64_t stride_p, void* kl,
                         int64_t B, int64_t K, int dtype, vae_stream_t stream) {
  if (B < 0 || K <= 0 || K > VAE_MAX_ROW) return VAE_ERR_ARG;
  if (B == 0) return VAE_OK;
  if (!logits_q || !kl || stride_q < K || (logits_p && stride_p < K)) return VAE_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == VAE_F32)
    return bern_kl_dispatch<float, false>(logits_q, stride_q, logits_p, stride_p, nullptr, kl, nullptr, nullptr, B, K, st);
  if (dtype == VAE_BF16)
    return bern_kl_dispatch<bf16_t, false>(logits_q, stride_q, logits_p, stride_p, nullptr, kl, nullptr, nullptr, B, K, st);
  if (dtype == VAE_F16)
    return bern_kl_dispatch<f16_t, false>(logits_q, stride_q, logits_p, stride_p, nullptr, kl, nullptr, nullptr, B, K, st);
  return VAE_ERR_ARG;
}

int vae_bernoulli_kl_bwd(const void* logits_q, int64_t stride_q, const void* logits_p, int64_t stride_p,
                         const void* gkl, void* g_logits_q, void* g_logits_p, int64_t B, int64_t K, int dtype,
                         vae_stream_t stream) {
  if (B < 0 || K <= 0 || K > VAE_MAX_ROW) return VAE_ERR_ARG;
  if (B == 0) return VAE_OK;
  if (!logits_q || !gkl || !g_logits_q || stride_q < K || (logits_p && stride_p < K)) return VAE_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == VAE_F32)
    return bern_kl_dispatch<float, true>(logits_q, stride_q, logits_p, stride_p, gkl, nullptr, g_logits_q, g_logits_p, B, K, st);
  if (dtype == VAE_BF16)
    return bern_kl_dispatch<bf16_t, true>(logits_q, stride_q, logits_p, stride_p, gkl, nullptr, g_logits_q, g_logits_p, B, K, st);
  if (dtype == VAE_F16)
    return bern_kl_dispatch<f16_t, true>(logits_q, stride_q, logits_p, stride_p, gkl, nullptr, g_logits_q, g_logits_p, B, K, st);
  return VAE_ERR_ARG;
}

}  // extern "C"
