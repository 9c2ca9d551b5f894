// One Gumbel-softmax row held in registers by a group of G lanes: shared by gumbel.hip (the stand-alone kernel)
// and concat.hip (the whole Concat(Gaussian, Gumbel) row in one launch).  Reference: reparameterizers/gumbel.py
// :119 (log_q_z), :269-277 (noise, temperature softmax), :296-301 (hard mask, straight-through sample).
#pragma once
#include "common.hpp"

namespace vae {

constexpr float kGumbelEps = 1e-9f;
constexpr float kNegInf = -3.402823466e+38f;

// l: the row's logits (lane `sub` of the group owns vectors sub, sub + G, ...; entries past the row end are clamped
// copies, masked here).  u: the injected uniforms when INJ, else scratch that receives the Philox draws keyed by the
// element index elem0 + j*V.  y / zh / lq: this ROW's output pointers, each may be null (kernel-uniform); nothing is
// stored unless `live`.  Every lane of the group (and, for G = kBlock, of the block) must call it.
// The arithmetic carries no branches: chunks past the row end are computed on their clamped copies and masked out
// of the reductions and the stores (a branch per chunk cost ~60 scalar-register spills and their wait states in
// the Concat row kernel, r03).
template <typename T, int V, int U, bool INJ>
__device__ __forceinline__ void gumbel_row_fwd(const float (&l)[U][V], float (&u)[U][V], T* __restrict__ y,
                                               T* __restrict__ zh, T* __restrict__ lq, bool live, bool nts, int sub,
                                               int G, int nvec, float tau, uint64_t seed, uint64_t offset,
                                               uint64_t elem0) {
  // Each output may be absent: the module classes write only what the step reads -- y in training -- and produce
  // z_hard / log_q_z on first read (vae_amd/lazy.py).
  const bool want_lq = lq != nullptr;
  const bool want_soft = y != nullptr || zh != nullptr;
  bool valid[U];
#pragma unroll
  for (int k = 0; k < U; ++k) valid[k] = sub + k * G < nvec;

  // ---- log_q_z = (l - max) - log(sum exp(l - max))
  float m1 = kNegInf, lse = 0.f;
  if (want_lq) {
#pragma unroll
    for (int k = 0; k < U; ++k)
#pragma unroll
      for (int i = 0; i < V; ++i) m1 = fmaxf(m1, valid[k] ? l[k][i] : kNegInf);
    m1 = row_max(m1, G);
    float s1 = 0.f;
#pragma unroll
    for (int k = 0; k < U; ++k)
#pragma unroll
      for (int i = 0; i < V; ++i) {
        const float ev = exp_acc(sub_rn(l[k][i], m1));
        s1 += valid[k] ? ev : 0.f;
      }
    s1 = row_sum(s1, G);
    lse = logf(s1);
  }
  if (!want_soft) {
#pragma unroll
    for (int k = 0; k < U; ++k) {
      if (live && valid[k]) {
        float o[V];
#pragma unroll
        for (int i = 0; i < V; ++i) o[i] = sub_rn(sub_rn(l[k][i], m1), lse);
        store_vec<T, V>(nts, lq + (int64_t)(sub + k * G) * V, o);
      }
    }
    return;
  }

  // ---- x = (l + g)/tau + eps, g = -log(-log(U + eps) + eps): every op rounds on its own
  float x[U][V];
  float m2 = kNegInf;
  const float itau = 1.0f / tau;
#pragma unroll
  for (int k = 0; k < U; ++k) {
    const int j = valid[k] ? sub + k * G : nvec - 1;
    if constexpr (!INJ) philox_uniform<V>(seed, offset, elem0 + (uint64_t)j * V, u[k]);
#pragma unroll
    for (int i = 0; i < V; ++i) {
      // injected U: the reference's own roundings (accurate logs, an IEEE divide, ocml expf below).  In-kernel
      // draws have nothing to be bit-compared with and that path is ALU-bound (Philox + three transcendentals per
      // element against 8 bytes of traffic): bare v_log_f32 (arguments >= 1e-9: normal range), a multiply by
      // 1/tau instead of the divide (~10 VALU ops), bare v_exp_f32 instead of expf (14).
      float g, xi;
      if constexpr (INJ) {
        g = -logf(add_rn(u[k][i], kGumbelEps));
        g = -logf(add_rn(g, kGumbelEps));
        g = to_f32(from_f32<T>(g));
        xi = to_f32(from_f32<T>(div_rn(to_f32(from_f32<T>(add_rn(l[k][i], g))), tau)));
      } else {
        g = -fast_log(u[k][i] + kGumbelEps);
        g = -fast_log(g + kGumbelEps);
        g = to_f32(from_f32<T>(g));
        xi = to_f32(from_f32<T>(mul_rn(to_f32(from_f32<T>(add_rn(l[k][i], g))), itau)));
      }
      x[k][i] = to_f32(from_f32<T>(add_rn(xi, kGumbelEps)));
      m2 = fmaxf(m2, valid[k] ? x[k][i] : kNegInf);
    }
  }
  m2 = row_max(m2, G);
  float s2 = 0.f;
#pragma unroll
  for (int k = 0; k < U; ++k)
#pragma unroll
    for (int i = 0; i < V; ++i) {
      if constexpr (INJ) x[k][i] = expf(sub_rn(x[k][i], m2));
      else x[k][i] = fast_exp(sub_rn(x[k][i], m2));
      s2 += valid[k] ? x[k][i] : 0.f;
    }
  s2 = row_sum(s2, G);
  const float rinv = div_rn(1.0f, s2);  // ATen's CPU softmax multiplies by the reciprocal
  float ymax = 0.f;
#pragma unroll
  for (int k = 0; k < U; ++k)
#pragma unroll
    for (int i = 0; i < V; ++i) {
      x[k][i] = to_f32(from_f32<T>(mul_rn(x[k][i], rinv)));
      ymax = fmaxf(ymax, valid[k] ? x[k][i] : 0.f);
    }
  // the largest y needs no reduction when z_hard is not wanted
  if (zh) ymax = row_max(ymax, G);

#pragma unroll
  for (int k = 0; k < U; ++k) {
    if (live && valid[k]) {
      const int64_t o0 = (int64_t)(sub + k * G) * V;
      if (y) store_vec<T, V>(nts, y + o0, x[k]);
      if (zh) {
        float h[V];
#pragma unroll
        for (int i = 0; i < V; ++i) {
          const float hot = (x[k][i] == ymax) ? 1.0f : 0.0f;
          h[i] = add_rn(to_f32(from_f32<T>(sub_rn(hot, x[k][i]))), x[k][i]);  // (h - y).detach() + y
        }
        store_vec<T, V>(nts, zh + o0, h);
      }
      if (want_lq) {
        float o[V];
#pragma unroll
        for (int i = 0; i < V; ++i) o[i] = sub_rn(sub_rn(l[k][i], m1), lse);
        store_vec<T, V>(nts, lq + o0, o);
      }
    }
  }
}

}  // namespace vae
