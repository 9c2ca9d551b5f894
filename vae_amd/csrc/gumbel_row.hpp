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
// EXACT: the caller guarantees nvec == U * G (every chunk of every lane is inside the row): no masks at all.
// The arithmetic carries no branches: chunks past the row end are computed on their clamped copies and masked out
// of the reductions and the stores (a branch per chunk cost ~60 scalar-register spills and their wait states in
// the Concat row kernel, r03).
template <typename T, int V, int U, bool INJ, bool EXACT = false>
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
  for (int k = 0; k < U; ++k) valid[k] = EXACT || sub + k * G < nvec;

  // ---- log_q_z = (l - max) - log(sum exp(l - max))
  float m1 = kNegInf, lse = 0.f;
  if (want_lq) {
#pragma unroll
    for (int k = 0; k < U; ++k)
#pragma unroll
      for (int i = 0; i < V; ++i) m1 = max_raw(m1, valid[k] ? l[k][i] : kNegInf);
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

  // ---- x = (l + g)/tau + eps, g = -log(-log(U + eps) + eps)
  float x[U][V];
  float m2 = kNegInf;
  float s2 = 0.f;
  if constexpr (INJ) {
    // injected U: the reference's own roundings, every op on its own (accurate logs, an IEEE divide, ocml expf),
    // through the tensor dtype wherever the reference holds a tensor
#pragma unroll
    for (int k = 0; k < U; ++k) {
#pragma unroll
      for (int i = 0; i < V; ++i) {
        float g = -logf(add_rn(u[k][i], kGumbelEps));
        g = -logf(add_rn(g, kGumbelEps));
        g = to_f32(from_f32<T>(g));
        const float xi = to_f32(from_f32<T>(div_rn(to_f32(from_f32<T>(add_rn(l[k][i], g))), tau)));
        x[k][i] = to_f32(from_f32<T>(add_rn(xi, kGumbelEps)));
        m2 = max_raw(m2, valid[k] ? x[k][i] : kNegInf);
      }
    }
    m2 = row_max(m2, G);
#pragma unroll
    for (int k = 0; k < U; ++k)
#pragma unroll
      for (int i = 0; i < V; ++i) {
        x[k][i] = expf(sub_rn(x[k][i], m2));
        s2 += valid[k] ? x[k][i] : 0.f;
      }
  } else {
    // In-kernel draws have nothing to be bit-compared with, and this path is VALU-bound (r03, rocprofv3: the Concat row
    // kernel issues ~700 VALU instructions per wave for 12 elements per lane; Philox is a third of them).  The same
    // formula in fp32 with the constants folded, in the log2 domain: with t = log2,
    //   u + eps       = fma(word >> 8, 2^-24, 2^-25 + eps)
    //   -ln(u+eps)+eps = fma(-ln2, t(u + eps), eps)                       =: g1
    //   log2(e) * x   = fma(t(g1), -ln2 * c, fma(l, c, eps * log2(e))),   c = log2(e) / tau
    // and y = exp2(that - rowmax) / sum: 10 VALU + 3 transcendental instructions per element.
    const float c = kLog2e / tau;
    const float c0 = kGumbelEps * kLog2e, c2 = -kLn2 * c;
#pragma unroll
    for (int k = 0; k < U; ++k) {
      const int j = valid[k] ? sub + k * G : nvec - 1;
      uint32_t w[V];
      philox_words<V>(seed, offset, elem0 + (uint64_t)j * V, w);
#pragma unroll
      for (int i = 0; i < V; ++i) {
        const float ue = __builtin_fmaf((float)(w[i] >> 8), 1.0f / 16777216.0f, 0.5f / 16777216.0f + kGumbelEps);
        const float g1 = __builtin_fmaf(-kLn2, log2_raw(ue), kGumbelEps);
        x[k][i] = __builtin_fmaf(log2_raw(g1), c2, __builtin_fmaf(l[k][i], c, c0));
        m2 = max_raw(m2, valid[k] ? x[k][i] : kNegInf);
      }
    }
    m2 = row_max(m2, G);
#pragma unroll
    for (int k = 0; k < U; ++k)
#pragma unroll
      for (int i = 0; i < V; ++i) {
        x[k][i] = __builtin_amdgcn_exp2f(x[k][i] - m2);
        s2 += valid[k] ? x[k][i] : 0.f;
      }
  }
  s2 = row_sum(s2, G);
  const float rinv = div_rn(1.0f, s2);  // ATen's CPU softmax multiplies by the reciprocal
  float ymax = 0.f;
#pragma unroll
  for (int k = 0; k < U; ++k)
#pragma unroll
    for (int i = 0; i < V; ++i) {
      x[k][i] = to_f32(from_f32<T>(mul_rn(x[k][i], rinv)));
      ymax = max_raw(ymax, valid[k] ? x[k][i] : 0.f);
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
