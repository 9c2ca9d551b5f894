// VRNN per-time-step pair: posterior + variance-clamped prior reparameterization and both KLs, fused.
//
// Replaces, per step, the op chain of /root/reference vrnn.py:419-447 (reparameterize):
//   :429      z_q, params_q = reparameterizer(encoder_logits)           (isotropic_gaussian.py:77-131, 204-215)
//   :432      prior_logits = cat([l[:, :L], sigmoid(l[:, L:])], -1)     (_clamp_variance, vrnn.py:395-408)
//   :435      z_p, params_p = reparameterizer(prior_logits)
// and of vrnn.py:723-733 (kld):
//   KL(N(mu_q, lv_q) || N(mu_p, lv_p)) [+ KL(N(mu_p, lv_p) || N(0,1)) when use_prior_kl]
//   (isotropic_gaussian.py:172-190, torch/distributions/kl.py:468-471; scales used as given).
// ~5 launches + 2 temporaries per step become one: read 16L (+8L noise), write 8L (z) + 8L (logvar)
// + 8L (clamped prior logits) + 8..12 bytes per sample.
#include "common.hpp"
#include "../../include/vae_amd.h"

namespace vae {

constexpr uint64_t kPriorStream = 0x8000000000000000ull;  // Philox stream id of the prior's draw

__device__ __forceinline__ float sigmoid_p(float v) { return rcp_nr(1.0f + exp_acc(-v)); }

template <typename T, int V, int U>
__global__ __launch_bounds__(kBlock) void gauss_pair_fwd_kernel(
    const T* __restrict__ enc, int64_t stride_e, const T* __restrict__ pri, int64_t stride_p,
    const T* __restrict__ noise_q, const T* __restrict__ noise_p, T* __restrict__ z_q, T* __restrict__ z_p,
    T* __restrict__ lv_q, T* __restrict__ lv_p, T* __restrict__ clamped, T* __restrict__ kl_pair,
    T* __restrict__ kl_prior, float* __restrict__ ws, float mean_scale, float* __restrict__ means, int64_t B, int L,
    float eps_tol, int training, uint64_t seed, uint64_t offset, int64_t row0, int G, int64_t tiles) {
  offset = resolve_offset(offset);
  const int rpb = kBlock / G;
  const bool nts = nt_stores((int64_t)B * L * (int64_t)sizeof(T));
  const int sub = threadIdx.x % G;
  const int rloc = threadIdx.x / G;
  const int nvec = L / V;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};  // sum mu_q, lv_q, mu_p, lv_p

  for (int64_t t = blockIdx.x; t < tiles; t += gridDim.x) {
    const int64_t row = t * rpb + rloc;
    float kp = 0.f, k0 = 0.f;
    if (row < B) {
      const T* er = enc + row * stride_e;
      const T* pr = pri + row * stride_p;
      for (int j0 = sub; j0 < nvec; j0 += G * U) {
        float mq[U][V], sq[U][V], mp[U][V], sp[U][V], eq[U][V], ep[U][V];
        load_chunk<T, V, U>(er, j0, G, nvec, mq);
        load_chunk<T, V, U>(er + L, j0, G, nvec, sq);
        load_chunk<T, V, U>(pr, j0, G, nvec, mp);
        load_chunk<T, V, U>(pr + L, j0, G, nvec, sp);
        if (training && noise_q) load_chunk<T, V, U>(noise_q + row * L, j0, G, nvec, eq);
        if (training && noise_p) load_chunk<T, V, U>(noise_p + row * L, j0, G, nvec, ep);
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int j = j0 + u * G;
          if (j < nvec) {
            const uint64_t elem = (uint64_t)(row0 + row) * (uint64_t)L + (uint64_t)j * V;
            if (training && !noise_q) philox_normal<V>(seed, offset, elem, eq[u]);
            if (training && !noise_p) philox_normal<V>(seed, offset ^ kPriorStream, elem, ep[u]);
            float zq[V], zp[V];
#pragma unroll
            for (int i = 0; i < V; ++i) {
              const float lq = to_f32(from_f32<T>(add_rn(sq[u][i], eps_tol)));
              const float cl = to_f32(from_f32<T>(sigmoid_p(sp[u][i])));  // the clamped prior "logvar" logit
              const float lp = to_f32(from_f32<T>(add_rn(cl, eps_tol)));
              const float a = mq[u][i], b = mp[u][i];
              sq[u][i] = lq;
              sp[u][i] = cl;
              if (training) {
                zq[i] = add_rn(to_f32(from_f32<T>(mul_rn(eq[u][i], to_f32(from_f32<T>(mul_rn(0.5f, lq)))))), a);
                zp[i] = add_rn(to_f32(from_f32<T>(mul_rn(ep[u][i], to_f32(from_f32<T>(mul_rn(0.5f, lp)))))), b);
              }
              // torch kl_normal_normal: var_ratio = (lq/lp)^2, t1 = ((a - b)/lp)^2
              const float ip = rcp_nr(lp);
              const float r = lq * ip;
              const float vr = r * r;
              const float d = (a - b) * ip;
              kp += 0.5f * (vr + d * d - 1.0f - fast_log(vr));
              const float lp2 = lp * lp;
              k0 += 0.5f * (lp2 + b * b - 1.0f - fast_log(lp2));
              acc[0] += a; acc[1] += lq; acc[2] += b; acc[3] += lp;
              eq[u][i] = lp;  // reuse as the logvar_prior store buffer
            }
            const int64_t o = row * L + (int64_t)j * V;
            if (training) {
              store_vec<T, V>(nts, z_q + o, zq);
              store_vec<T, V>(nts, z_p + o, zp);
            }
            if (lv_q) store_vec<T, V>(nts, lv_q + o, sq[u]);
            if (lv_p) store_vec<T, V>(nts, lv_p + o, eq[u]);
            store_vec<T, V>(nts, clamped + row * 2 * (int64_t)L + (int64_t)j * V, mp[u]);
            store_vec<T, V>(nts, clamped + row * 2 * (int64_t)L + L + (int64_t)j * V, sp[u]);
          }
        }
      }
    }
    kp = group_sum(kp, G);
    k0 = group_sum(k0, G);
    if (row < B && sub == 0) {
      kl_pair[row] = from_f32<T>(kp);
      if (kl_prior) kl_prior[row] = from_f32<T>(k0);
    }
  }
  if (ws) grid_reduce<4>(acc, ws, mean_scale, means, kNanGauss2 | (kNanGauss2 << 8));
}

template <typename T, int V, int U>
__global__ __launch_bounds__(kBlock) void gauss_pair_bwd_kernel(
    const T* __restrict__ enc, int64_t stride_e, const T* __restrict__ pri, int64_t stride_p,
    const T* __restrict__ noise_q, const T* __restrict__ noise_p, const T* __restrict__ gz_q,
    const T* __restrict__ gz_p, const T* __restrict__ glv_q, const T* __restrict__ glv_p,
    const T* __restrict__ gclamped, const T* __restrict__ gkl_pair, const T* __restrict__ gkl_prior,
    const float* __restrict__ gmeans, T* __restrict__ g_enc, T* __restrict__ g_pri, int64_t B, int L, float eps_tol,
    float mean_scale, int training, uint64_t seed, uint64_t offset, int64_t row0, int G, int64_t tiles) {
  offset = resolve_offset(offset);
  const int rpb = kBlock / G;
  const bool nts = nt_stores((int64_t)B * L * (int64_t)sizeof(T));
  const int sub = threadIdx.x % G;
  const int rloc = threadIdx.x / G;
  const int nvec = L / V;
  float gm[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) gm[k] = gmeans ? gmeans[k] * mean_scale : 0.f;
  const bool use_eq = training && gz_q, use_ep = training && gz_p;

  for (int64_t t = blockIdx.x; t < tiles; t += gridDim.x) {
    const int64_t row = t * rpb + rloc;
    if (row >= B) continue;
    const T* er = enc + row * stride_e;
    const T* pr = pri + row * stride_p;
    const float gk = gkl_pair ? to_f32(gkl_pair[row]) : 0.f;
    const float g0 = gkl_prior ? to_f32(gkl_prior[row]) : 0.f;
    for (int j0 = sub; j0 < nvec; j0 += G * U) {
      float mq[U][V], sq[U][V], mp[U][V], sp[U][V], eq[U][V], ep[U][V], a1[U][V], a2[U][V];
      load_chunk<T, V, U>(er, j0, G, nvec, mq);
      load_chunk<T, V, U>(er + L, j0, G, nvec, sq);
      load_chunk<T, V, U>(pr, j0, G, nvec, mp);
      load_chunk<T, V, U>(pr + L, j0, G, nvec, sp);
      if (use_eq && noise_q) load_chunk<T, V, U>(noise_q + row * L, j0, G, nvec, eq);
      if (use_ep && noise_p) load_chunk<T, V, U>(noise_p + row * L, j0, G, nvec, ep);
      if (gz_q) load_chunk<T, V, U>(gz_q + row * L, j0, G, nvec, a1);
      if (gz_p) load_chunk<T, V, U>(gz_p + row * L, j0, G, nvec, a2);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int j = j0 + u * G;
        if (j < nvec) {
          const uint64_t elem = (uint64_t)(row0 + row) * (uint64_t)L + (uint64_t)j * V;
          if (use_eq && !noise_q) philox_normal<V>(seed, offset, elem, eq[u]);
          if (use_ep && !noise_p) philox_normal<V>(seed, offset ^ kPriorStream, elem, ep[u]);
          // the rarely used cotangents are read only when present
          float x1[V], x2[V], c1[V], c2[V];
          const int64_t o = row * L + (int64_t)j * V;
          if (glv_q) load_vec<T, V>(glv_q + o, x1);
          if (glv_p) load_vec<T, V>(glv_p + o, x2);
          if (gclamped) {
            load_vec<T, V>(gclamped + row * 2 * (int64_t)L + (int64_t)j * V, c1);
            load_vec<T, V>(gclamped + row * 2 * (int64_t)L + L + (int64_t)j * V, c2);
          }
          float omq[V], osq[V], omp[V], osp[V];
#pragma unroll
          for (int i = 0; i < V; ++i) {
            const float lq = to_f32(from_f32<T>(add_rn(sq[u][i], eps_tol)));
            const float cl = to_f32(from_f32<T>(sigmoid_p(sp[u][i])));
            const float lp = to_f32(from_f32<T>(add_rn(cl, eps_tol)));
            const float a = mq[u][i], b = mp[u][i];
            const float ip = fast_rcp(lp), ip2 = ip * ip;
            const float d = a - b;
            const float gzq = gz_q ? a1[u][i] : 0.f, gzp = gz_p ? a2[u][i] : 0.f;
            const float pm = gk * d * ip2;  // d kl_pair / d mu_q = -d kl_pair / d mu_p
            omq[i] = gzq + pm + gm[0];
            osq[i] = (use_eq ? 0.5f * eq[u][i] * gzq : 0.f) + gk * (lq * ip2 - fast_rcp(lq)) + gm[1] +
                     (glv_q ? x1[i] : 0.f);
            omp[i] = gzp - pm + g0 * b + gm[2] + (gclamped ? c1[i] : 0.f);
            const float glp = (use_ep ? 0.5f * ep[u][i] * gzp : 0.f) + gk * (ip - (lq * lq + d * d) * ip2 * ip) +
                              g0 * (lp - ip) + gm[3] + (glv_p ? x2[i] : 0.f);
            osp[i] = (glp + (gclamped ? c2[i] : 0.f)) * cl * (1.0f - cl);
          }
          store_vec<T, V>(nts, g_enc + row * 2 * (int64_t)L + (int64_t)j * V, omq);
          store_vec<T, V>(nts, g_enc + row * 2 * (int64_t)L + L + (int64_t)j * V, osq);
          store_vec<T, V>(nts, g_pri + row * 2 * (int64_t)L + (int64_t)j * V, omp);
          store_vec<T, V>(nts, g_pri + row * 2 * (int64_t)L + L + (int64_t)j * V, osp);
        }
      }
    }
  }
}

template <typename T>
static int pair_fwd_dispatch(const void* enc, int64_t se, const void* pri, int64_t sp, const void* nq, const void* np,
                             void* zq, void* zp, void* lvq, void* lvp, void* clamped, void* klp, void* kl0,
                             float* means, float* workspace, int64_t B, int64_t L, float eps_tol, float mean_scale,
                             int training, uint64_t seed, uint64_t offset, int64_t row0, hipStream_t stream) {
  constexpr int VN = VecOf<T>::n;
  const bool vec = (L % VN == 0) && (se % VN == 0) && (sp % VN == 0) && aligned16(enc) && aligned16(pri) &&
                   (!lvq || aligned16(lvq)) && (!lvp || aligned16(lvp)) && aligned16(clamped) && (!nq || aligned16(nq)) &&
                   (!np || aligned16(np)) && (!zq || aligned16(zq)) && (!zp || aligned16(zp));
  RowGeom g = row_geom(B, vec ? L / VN : L, means ? kGridReduce : kGridStream);
  {
    // r02 cold-cache sweep: the many live arrays of this kernel make one chunk in flight the better choice for short
    // rows (L = 128: 26.1 -> 23.5 us), two rows per wave for rows of 65..128 vectors (L = 512: 247 -> 238 us)
    const int64_t vecs = vec ? L / VN : L;
    if (vecs >= 16 && vecs <= 32) regroup(g, B, 16, 1, means ? kGridReduce : kGridStream);
    else if (vecs > 64 && vecs <= 128) regroup(g, B, 32, 2, means ? kGridReduce : kGridStream);
  }
  float* ws = means ? workspace : nullptr;
  if (vec) {
    VAE_SWITCH_U(g.U < 4 ? g.U : 2,
                 hipLaunchKernelGGL((gauss_pair_fwd_kernel<T, VN, U>), dim3(g.grid), dim3(kBlock), 0, stream,
                                    (const T*)enc, se, (const T*)pri, sp, (const T*)nq, (const T*)np, (T*)zq, (T*)zp,
                                    (T*)lvq, (T*)lvp, (T*)clamped, (T*)klp, (T*)kl0, ws, mean_scale, means, B, (int)L,
                                    eps_tol, training, seed, offset, row0, g.G, g.tiles));
  } else {
    VAE_SWITCH_U(g.U < 4 ? g.U : 2,
                 hipLaunchKernelGGL((gauss_pair_fwd_kernel<T, 1, U>), dim3(g.grid), dim3(kBlock), 0, stream,
                                    (const T*)enc, se, (const T*)pri, sp, (const T*)nq, (const T*)np, (T*)zq, (T*)zp,
                                    (T*)lvq, (T*)lvp, (T*)clamped, (T*)klp, (T*)kl0, ws, mean_scale, means, B, (int)L,
                                    eps_tol, training, seed, offset, row0, g.G, g.tiles));
  }
  return hipGetLastError() == hipSuccess ? VAE_OK : VAE_ERR_LAUNCH;
}

template <typename T>
static int pair_bwd_dispatch(const void* enc, int64_t se, const void* pri, int64_t sp, const void* nq, const void* np,
                             const void* gzq, const void* gzp, const void* glvq, const void* glvp, const void* gcl,
                             const void* gklp, const void* gkl0, const float* gmeans, void* genc, void* gpri,
                             int64_t B, int64_t L, float eps_tol, float mean_scale, int training, uint64_t seed,
                             uint64_t offset, int64_t row0, hipStream_t stream) {
  constexpr int VN = VecOf<T>::n;
  bool vec = (L % VN == 0) && (se % VN == 0) && (sp % VN == 0) && aligned16(enc) && aligned16(pri) &&
             aligned16(genc) && aligned16(gpri);
  const void* opt[] = {nq, np, gzq, gzp, glvq, glvp, gcl};
  for (const void* p : opt) vec = vec && (!p || aligned16(p));
  const RowGeom g = row_geom(B, vec ? L / VN : L, kGridStream);
  if (vec) {
    VAE_SWITCH_U(g.U < 4 ? g.U : 2,
                 hipLaunchKernelGGL((gauss_pair_bwd_kernel<T, VN, U>), dim3(g.grid), dim3(kBlock), 0, stream,
                                    (const T*)enc, se, (const T*)pri, sp, (const T*)nq, (const T*)np, (const T*)gzq,
                                    (const T*)gzp, (const T*)glvq, (const T*)glvp, (const T*)gcl, (const T*)gklp,
                                    (const T*)gkl0, gmeans, (T*)genc, (T*)gpri, B, (int)L, eps_tol, mean_scale,
                                    training, seed, offset, row0, g.G, g.tiles));
  } else {
    VAE_SWITCH_U(g.U < 4 ? g.U : 2,
                 hipLaunchKernelGGL((gauss_pair_bwd_kernel<T, 1, U>), dim3(g.grid), dim3(kBlock), 0, stream,
                                    (const T*)enc, se, (const T*)pri, sp, (const T*)nq, (const T*)np, (const T*)gzq,
                                    (const T*)gzp, (const T*)glvq, (const T*)glvp, (const T*)gcl, (const T*)gklp,
                                    (const T*)gkl0, gmeans, (T*)genc, (T*)gpri, B, (int)L, eps_tol, mean_scale,
                                    training, seed, offset, row0, g.G, g.tiles));
  }
  return hipGetLastError() == hipSuccess ? VAE_OK : VAE_ERR_LAUNCH;
}

}  // namespace vae

using namespace vae;

extern "C" {

int vae_gauss_pair_fwd(const void* enc_logits, int64_t stride_enc, const void* prior_logits, int64_t stride_prior,
                       const void* noise_post, const void* noise_prior, void* z_post, void* z_prior,
                       void* logvar_post, void* logvar_prior, void* clamped_prior_logits, void* kl_pair,
                       void* kl_prior, float* means, float* workspace, int64_t B, int64_t L, float eps_tol,
                       float mean_scale, int training, uint64_t seed, uint64_t offset, int64_t row0, int dtype,
                       vae_stream_t stream) {
  if (B < 0 || L <= 0 || L > VAE_MAX_ROW) return VAE_ERR_ARG;
  if (B == 0) return VAE_OK;
  if (!enc_logits || !prior_logits || !clamped_prior_logits || !kl_pair ||
      (training && (!z_post || !z_prior)) || (means && !workspace) || stride_enc < 2 * L || stride_prior < 2 * L)
    return VAE_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == VAE_F32)
    return pair_fwd_dispatch<float>(enc_logits, stride_enc, prior_logits, stride_prior, noise_post, noise_prior,
                                    z_post, z_prior, logvar_post, logvar_prior, clamped_prior_logits, kl_pair, kl_prior,
                                    means, workspace, B, L, eps_tol, mean_scale, training, seed, offset, row0, st);
  if (dtype == VAE_BF16)
    return pair_fwd_dispatch<bf16_t>(enc_logits, stride_enc, prior_logits, stride_prior, noise_post, noise_prior,
                                     z_post, z_prior, logvar_post, logvar_prior, clamped_prior_logits, kl_pair,
                                     kl_prior, means, workspace, B, L, eps_tol, mean_scale, training, seed, offset,
                                     row0, st);
  if (dtype == VAE_F16)
    return pair_fwd_dispatch<f16_t>(enc_logits, stride_enc, prior_logits, stride_prior, noise_post, noise_prior,
                                     z_post, z_prior, logvar_post, logvar_prior, clamped_prior_logits, kl_pair,
                                     kl_prior, means, workspace, B, L, eps_tol, mean_scale, training, seed, offset,
                                     row0, st);
  return VAE_ERR_ARG;
}

int vae_gauss_pair_bwd(const void* enc_logits, int64_t stride_enc, const void* prior_logits, int64_t stride_prior,
                       const void* noise_post, const void* noise_prior, const void* gz_post, const void* gz_prior,
                       const void* glogvar_post, const void* glogvar_prior, const void* gclamped, const void* gkl_pair,
                       const void* gkl_prior, const float* gmeans, void* g_enc_logits, void* g_prior_logits,
                       int64_t B, int64_t L, float eps_tol, float mean_scale, int training, uint64_t seed,
                       uint64_t offset, int64_t row0, int dtype, vae_stream_t stream) {
  if (B < 0 || L <= 0 || L > VAE_MAX_ROW) return VAE_ERR_ARG;
  if (B == 0) return VAE_OK;
  if (!enc_logits || !prior_logits || !g_enc_logits || !g_prior_logits || stride_enc < 2 * L || stride_prior < 2 * L)
    return VAE_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == VAE_F32)
    return pair_bwd_dispatch<float>(enc_logits, stride_enc, prior_logits, stride_prior, noise_post, noise_prior,
                                    gz_post, gz_prior, glogvar_post, glogvar_prior, gclamped, gkl_pair, gkl_prior,
                                    gmeans, g_enc_logits, g_prior_logits, B, L, eps_tol, mean_scale, training, seed,
                                    offset, row0, st);
  if (dtype == VAE_BF16)
    return pair_bwd_dispatch<bf16_t>(enc_logits, stride_enc, prior_logits, stride_prior, noise_post, noise_prior,
                                     gz_post, gz_prior, glogvar_post, glogvar_prior, gclamped, gkl_pair, gkl_prior,
                                     gmeans, g_enc_logits, g_prior_logits, B, L, eps_tol, mean_scale, training, seed,
                                     offset, row0, st);
  if (dtype == VAE_F16)
    return pair_bwd_dispatch<f16_t>(enc_logits, stride_enc, prior_logits, stride_prior, noise_post, noise_prior,
                                     gz_post, gz_prior, glogvar_post, glogvar_prior, gclamped, gkl_pair, gkl_prior,
                                     gmeans, g_enc_logits, g_prior_logits, B, L, eps_tol, mean_scale, training, seed,
                                     offset, row0, st);
  return VAE_ERR_ARG;
}

}  // extern "C"
