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

// lgamma / digamma / trigamma for x > 0 from ONE recurrence: shift x up to y = x + n >= 6 (n <= 6;
// concentrations here are sigmoids, so x < 2.2 and n = 4..6), Stirling / asymptotic series at y, then
//   lgamma(x) = lgamma(y) - log(x (x+1) ... (x+n-1)),  psi(x) = psi(y) - sum 1/(x+k),  psi'(x) = psi'(y) + sum 1/(x+k)^2.
// ~35 instructions for all three, against ~100 for ocml's lgammaf alone (the fused KL needs 3 lgamma + 3 digamma
// per element: this kernel was ALU-bound at 0.9 TB/s with the library forms).  Absolute error ~1e-6.
template <bool LG, bool PSI, bool TRI>
__device__ __forceinline__ void gamma_fns(float x, float& lg, float& psi, float& tri) {
  float prod = 1.0f, r1 = 0.f, r2 = 0.f;
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    if (x < 6.0f) {
      const float ix = fast_rcp(x);
      if (LG) prod *= x;
      if (PSI) r1 += ix;
      if (TRI) r2 += ix * ix;
      x += 1.0f;
    }
  }
  const float i = fast_rcp(x), i2 = i * i;
  const float lx = fast_log(x);
  if (LG)
    lg = (x - 0.5f) * lx - x + 0.91893853320467274178f + i * (1.0f / 12.0f - i2 * (1.0f / 360.0f - i2 * (1.0f / 1260.0f))) -
         fast_log(prod);
  if (PSI) psi = lx - 0.5f * i - i2 * (1.0f / 12.0f - i2 * (1.0f / 120.0f - i2 * (1.0f / 252.0f))) - r1;
  if (TRI) tri = i + 0.5f * i2 + i * i2 * (1.0f / 6.0f - i2 * (1.0f / 30.0f - i2 * (1.0f / 42.0f))) + r2;
}

// The same three functions for 0 < x < 2.25 (a sigmoid concentration, or the sum of two): the shift is always
// n = 6, so the product P(x) = x (x+1) ... (x+5) and sum_k 1/(x+k) = P'/P, sum_k 1/(x+k)^2 = (P'/P)^2 - P''/P are
// fixed polynomials (all coefficients positive: no cancellation in P, P', P''), i.e. 15 fma + ONE reciprocal in
// place of six reciprocals and six compare/select groups.  lgamma is returned WITHOUT its -log P term; the caller
// merges the three logs of a KL element into one (log(P1 P0 / Ps)).
template <bool LG, bool PSI, bool TRI>
__device__ __forceinline__ void gamma_small(float x, float& lg_nolog, float& P, float& psi, float& tri) {
  P = (((((x + 15.0f) * x + 85.0f) * x + 225.0f) * x + 274.0f) * x + 120.0f) * x;
  const float y = x + 6.0f;
  const float i = fast_rcp(y), i2 = i * i;
  const float ly = fast_log(y);
  if (LG) lg_nolog = (y - 0.5f) * ly - y + 0.91893853320467274178f + i * (1.0f / 12.0f - i2 * (1.0f / 360.0f - i2 * (1.0f / 1260.0f)));
  if (PSI || TRI) {
    const float iP = fast_rcp(P);
    const float d1 = ((((6.0f * x + 75.0f) * x + 340.0f) * x + 675.0f) * x + 548.0f) * x + 120.0f;
    const float r1 = d1 * iP;
    if (PSI) psi = ly - 0.5f * i - i2 * (1.0f / 12.0f - i2 * (1.0f / 120.0f - i2 * (1.0f / 252.0f))) - r1;
    if (TRI) {
      const float d2 = (((30.0f * x + 300.0f) * x + 1020.0f) * x + 1350.0f) * x + 548.0f;
      tri = i + 0.5f * i2 + i * i2 * (1.0f / 6.0f - i2 * (1.0f / 30.0f - i2 * (1.0f / 42.0f))) + (r1 * r1 - d2 * iP);
    }
  }
}

// gamma_small<LG, PSI, false> on TWO arguments at once, laid out for the packed-fp32 pipe (v_pk_fma_f32 /
// v_pk_mul_f32 / v_pk_add_f32: two values per lane and instruction).  The forward KL is arithmetic-bound (r03:
// ~110 full-rate + 14 quarter-rate instructions per element pair, 152 us where the bytes need 88): the polynomials are
// two thirds of it and pack perfectly; the reciprocals and logs stay one per value.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 rcp2(f32x2 v) { return f32x2{fast_rcp(v[0]), fast_rcp(v[1])}; }
__device__ __forceinline__ f32x2 ln2_(f32x2 v) { return f32x2{log2_raw(v[0]), log2_raw(v[1])} * kLn2; }
__device__ __forceinline__ void gamma_small2(f32x2 x, f32x2& lg_nolog, f32x2& P, f32x2& psi) {
  P = (((((x + 15.0f) * x + 85.0f) * x + 225.0f) * x + 274.0f) * x + 120.0f) * x;
  const f32x2 y = x + 6.0f;
  const f32x2 i = rcp2(y), i2 = i * i;
  const f32x2 ly = ln2_(y);
  lg_nolog = (y - 0.5f) * ly - y + 0.91893853320467274178f + i * (1.0f / 12.0f - i2 * (1.0f / 360.0f - i2 * (1.0f / 1260.0f)));
  const f32x2 iP = rcp2(P);   // (one reciprocal of y P for both 1/y and 1/P: measured, no gain -- r02 and r03)
  const f32x2 d1 = ((((6.0f * x + 75.0f) * x + 340.0f) * x + 675.0f) * x + 548.0f) * x + 120.0f;
  psi = ly - 0.5f * i - i2 * (1.0f / 12.0f - i2 * (1.0f / 120.0f - i2 * (1.0f / 252.0f))) - d1 * iP;
}
// The KL of TWO elements (a1[e], a0[e]), e = 0, 1, against the fixed prior Beta(b, b): three packed evaluations --
// {a1[0], a0[0]}, {a1[1], a0[1]}, {sp[0], sp[1]} -- where beta_kl_elem_small makes six scalar ones.  Returns
// kl[0] + kl[1] (the caller sums over the row anyway).
__device__ __forceinline__ float beta_kl_pair_small(float a1x, float a0x, float a1y, float a0y, float b, float lgq) {
  const f32x2 ex = {a1x, a0x}, ey = {a1y, a0y}, sp = {a1x + a0x, a1y + a0y};
  f32x2 lx, Px, px, ly, Py, py, ls, Ps, ps;
  gamma_small2(ex, lx, Px, px);
  gamma_small2(ey, ly, Py, py);
  gamma_small2(sp, ls, Ps, ps);
  // lgamma(sp) - lgamma(a1) - lgamma(a0) = (ls - l1 - l0) + log(P1 P0 / Ps), per element
  const f32x2 l10 = {lx[0] + lx[1], ly[0] + ly[1]};
  const f32x2 pp = {Px[0] * Px[1], Py[0] * Py[1]};
  const f32x2 lgp = (ls - l10) + ln2_(pp * rcp2(Ps));
  const f32x2 dx = (ex - b) * px, dy = (ey - b) * py;          // {(a1 - b) psi(a1), (a0 - b) psi(a0)}
  const f32x2 tail = {dx[0] + dx[1], dy[0] + dy[1]};
  const f32x2 kl = lgq + lgp + tail + (2.0f * b - sp) * ps;
  return kl[0] + kl[1];
}

// KL(Beta(a1, a0) || Beta(b, b)) for sigmoid concentrations a1, a0 in (0, 1] against the fixed prior
__device__ __forceinline__ float beta_kl_elem_small(float a1, float a0, float b, float lgq) {
  const float sp = a1 + a0;
  float l1, l0, ls, P1, P0, Ps, p1, p0, ps, unused;
  gamma_small<true, true, false>(a1, l1, P1, p1, unused);
  gamma_small<true, true, false>(a0, l0, P0, p0, unused);
  gamma_small<true, true, false>(sp, ls, Ps, ps, unused);
  // lgamma(sp) - lgamma(a1) - lgamma(a0) = (ls - l1 - l0) + log(P1 P0 / Ps)
  const float lgp = (ls - l1 - l0) + fast_log(P1 * P0 * fast_rcp(Ps));
  return lgq + lgp + (a1 - b) * p1 + (a0 - b) * p0 + (2.0f * b - sp) * ps;
}
__device__ __forceinline__ void beta_kl_grad_small(float a1, float a0, float b, float& ga1, float& ga0) {
  const float sp = a1 + a0;
  float u, P, p, t1, t0, ts;
  gamma_small<false, false, true>(a1, u, P, p, t1);
  gamma_small<false, false, true>(a0, u, P, p, t0);
  gamma_small<false, false, true>(sp, u, P, p, ts);
  const float tsp = (2.0f * b - sp) * ts;
  ga1 = (a1 - b) * t1 + tsp;
  ga0 = (a0 - b) * t0 + tsp;
}

// KL(Beta(a1, a0) || Beta(b1, b0)); lgq = lgamma(b1) + lgamma(b0) - lgamma(b1 + b0) of the prior
__device__ __forceinline__ float beta_kl_elem(float a1, float a0, float b1, float b0, float lgq) {
  const float sp = a1 + a0, sq = b1 + b0;
  float l1, l0, ls, p1, p0, ps, unused;
  gamma_fns<true, true, false>(a1, l1, p1, unused);
  gamma_fns<true, true, false>(a0, l0, p0, unused);
  gamma_fns<true, true, false>(sp, ls, ps, unused);
  return lgq + ls - l1 - l0 + (a1 - b1) * p1 + (a0 - b0) * p0 + (sq - sp) * ps;
}
__device__ __forceinline__ float beta_prior_lg(float b1, float b0) {
  float l1, l0, ls, u1, u2;
  gamma_fns<true, false, false>(b1, l1, u1, u2);
  gamma_fns<true, false, false>(b0, l0, u1, u2);
  gamma_fns<true, false, false>(b1 + b0, ls, u1, u2);
  return l1 + l0 - ls;
}
// lgamma(1.1) + lgamma(1.1) - lgamma(2.2) for the fixed Beta(1.1, 1.1) prior
constexpr float kPriorLg = 2.0f * -0.04987244125983972f - 0.09694746679063826f;

// d kl / d(a1, a0) and (optionally) d kl / d(b1, b0)
__device__ __forceinline__ void beta_kl_grad(float a1, float a0, float b1, float b0, float& ga1, float& ga0,
                                             float& gb1, float& gb0, bool prior_grads) {
  const float sp = a1 + a0, sq = b1 + b0;
  float u, p1, p0, ps, t1, t0, ts;
  gamma_fns<false, true, true>(a1, u, p1, t1);
  gamma_fns<false, true, true>(a0, u, p0, t0);
  gamma_fns<false, true, true>(sp, u, ps, ts);
  const float tsp = (sq - sp) * ts;
  ga1 = (a1 - b1) * t1 + tsp;
  ga0 = (a0 - b0) * t0 + tsp;
  if (prior_grads) {
    float q1, q0, qs, v;
    gamma_fns<false, true, false>(b1, u, q1, v);
    gamma_fns<false, true, false>(b0, u, q0, v);
    gamma_fns<false, true, false>(sq, u, qs, v);
    gb1 = q1 - qs - p1 + ps;
    gb0 = q0 - qs - p0 + ps;
  }
}
// v_exp_f32 + Newton-refined reciprocal: ~2 ulp on the concentration (tolerance rtol 1e-5), a third of the ocml expf form
__device__ __forceinline__ float sigmoid_b(float v) { return rcp_nr(1.0f + fast_exp(-v)); }

// ---- fused: logits [B, 2L] -> conc1, conc0 [B, L], kl [B] vs Beta(1.1, 1.1), partial sums of conc
template <typename T, int V, int U>
__global__ __launch_bounds__(kBlock) void beta_split_fwd_kernel(
    const T* __restrict__ logits, int64_t row_stride, T* __restrict__ c1, T* __restrict__ c0, T* __restrict__ kl,
    float* __restrict__ ws, float mean_scale, float* __restrict__ means, int64_t B, int L, int G, int64_t tiles) {
  const int rpb = kBlock / G;
  const bool nts = nt_stores((int64_t)B * L * (int64_t)sizeof(T));
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
              acc[0] += a[k][i];
              acc[1] += b[k][i];
            }
            if constexpr (V % 2 == 0) {
#pragma unroll
              for (int i = 0; i < V; i += 2)
                klr += beta_kl_pair_small(a[k][i], b[k][i], a[k][i + 1], b[k][i + 1], kBetaPrior, kPriorLg);
            } else {
#pragma unroll
              for (int i = 0; i < V; ++i) klr += beta_kl_elem_small(a[k][i], b[k][i], kBetaPrior, kPriorLg);
            }
            store_vec<T, V>(nts, c1 + row * L + (int64_t)j * V, a[k]);
            store_vec<T, V>(nts, c0 + row * L + (int64_t)j * V, b[k]);
          }
        }
      }
    }
    klr = group_sum(klr, G);
    if (row < B && sub == 0) kl[row] = from_f32<T>(klr);
  }
  if (ws) grid_reduce<2>(acc, ws, mean_scale, means, kNanGauss2);
}

// glogits = sigmoid' * (g_conc + gkl * dKL/dconc + g_mean * mean_scale)
template <typename T, int V, int U>
__global__ __launch_bounds__(kBlock) void beta_split_bwd_kernel(
    const T* __restrict__ logits, int64_t row_stride, const T* __restrict__ gc1, const T* __restrict__ gc0,
    const T* __restrict__ gkl, const float* __restrict__ gmeans, T* __restrict__ glogits, int64_t B, int L,
    float mean_scale, int G, int64_t tiles) {
  const int rpb = kBlock / G;
  const bool nts = nt_stores((int64_t)B * L * (int64_t)sizeof(T));
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
            float d1 = 0.f, d0 = 0.f;
            if (gkl) beta_kl_grad_small(s1, s0, kBetaPrior, d1, d0);
            o1[i] = s1 * (1.0f - s1) * ((gc1 ? g1[k][i] : 0.f) + gk * d1 + gm1);
            o0[i] = s0 * (1.0f - s0) * ((gc0 ? g0[k][i] : 0.f) + gk * d0 + gm0);
          }
          store_vec<T, V>(nts, grow + (int64_t)j * V, o1);
          store_vec<T, V>(nts, grow + L + (int64_t)j * V, o0);
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
  const bool nts = nt_stores((int64_t)B * L * (int64_t)sizeof(T));
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
                acc += beta_kl_elem(x1[k][i], x0[k][i], q1, q0, b1 ? beta_prior_lg(q1, q0) : kPriorLg);
              } else {
                beta_kl_grad(x1[k][i], x0[k][i], q1, q0, o1[i], o0[i], p1[i], p0[i], gb1 != nullptr);
                o1[i] *= gk; o0[i] *= gk;
                if (gb1) { p1[i] *= gk; p0[i] *= gk; }
              }
            }
            if constexpr (BWD) {
              const int64_t o = row * L + (int64_t)j * V;
              store_vec<T, V>(nts, ga1 + o, o1);
              store_vec<T, V>(nts, ga0 + o, o0);
              if (gb1) {
                store_vec<T, V>(nts, gb1 + o, p1);
                store_vec<T, V>(nts, gb0 + o, p0);
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
  if (dtype == VAE_F16)
    return beta_split_fwd_dispatch<f16_t>(logits, row_stride, conc1, conc2, kl, means, workspace, B, L, mean_scale, st);
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
  if (dtype == VAE_F16)
    return beta_split_bwd_dispatch<f16_t>(logits, row_stride, gconc1, gconc2, gkl, gmeans, glogits, B, L, mean_scale, st);
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
  if (dtype == VAE_F16)
    return beta_kl_dispatch<f16_t, false>(a1, stride_a1, a0, stride_a0, b1, stride_b1, b0, stride_b0, nullptr, kl,
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
  if (dtype == VAE_F16)
    return beta_kl_dispatch<f16_t, true>(a1, stride_a1, a0, stride_a0, b1, stride_b1, b0, stride_b0, gkl, nullptr,
                                          ga1, ga0, gb1, gb0, B, L, st);
  return VAE_ERR_ARG;
}

}  // extern "C"
