// IsotropicGaussian reparameterize + KL, fused (forward and backward).
//
// Replaces the op chain of /root/reference reparameterizers/isotropic_gaussian.py:
//   :123-125  mu = logits[..., :L]; s = logits[..., L:] + eps_tol
//   :86-93    std = 0.5*s (NO exp); e ~ N(0,1); z = e*std + mu      (train / force)
//   :96       z = mu                                               (eval)
//   :168-170  KL(N(mu,s) || N(0,1)) = sum_j 0.5*(s^2 + mu^2 - 1 - log(s^2))
//   :186-190  KL(N(mu_q,s_q) || N(mu_p,s_p)) (torch/distributions/kl.py:468-471)
//   :213-214  mu_mean, logvar_mean
// and of vrnn.py:395-417 (_clamp_variance: s_p = sigmoid(raw)) for the pair form.
//
// HBM-bound: per sample the forward reads 8L (logits) [+4L noise], writes 4L (z)
// [+4L logvar] + 4 (kl).  One group of G lanes per row, 16 B per lane per access.
#include "common.hpp"
#include "../../include/vae_amd.h"

namespace vae {

#ifndef VAE_NT_GAUSS_FWD
#define VAE_NT_GAUSS_FWD 1
#endif
#ifndef VAE_NT_GAUSS_BWD
#define VAE_NT_GAUSS_BWD 1
#endif
constexpr bool kNtGaussFwd = VAE_NT_GAUSS_FWD != 0, kNtGaussBwd = VAE_NT_GAUSS_BWD != 0;

// Where a row's mu and s vectors live.  group == 0: the [B, 2L] layout (mu at row * row_stride, s right behind it at
// + L).  group > 0: 4-D logits [B, 2C, H, W] split on the CHANNEL dimension (isotropic_gaussian.py:108-114) with the
// reduction over W only: the rows are (b, c, h), L = W, `group` = C*H rows of stride row_stride = W per sample, samples
// `group_stride` = 2*C*H*W apart, and a row's s vector s_off = C*H*W elements behind its mu vector.
struct RowGroups {
  int64_t group, group_stride, s_off;
  __device__ __forceinline__ int64_t offset(int64_t row, int64_t row_stride) const {
    if (!group) return row * row_stride;
    const int64_t b = row / group;
    return b * group_stride + (row - b * group) * row_stride;
  }
};

// ---------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------
template <typename T, int V, int U>
__global__ __launch_bounds__(kBlock) void gauss_fwd_kernel(
    const T* __restrict__ logits, int64_t row_stride, const T* __restrict__ noise, T* __restrict__ z,
    int64_t z_stride, T* __restrict__ s_out, T* __restrict__ kl, float* __restrict__ ws, float mean_scale,
    float* __restrict__ means, int64_t B, int L, float eps_tol, int training, uint64_t seed, uint64_t offset, int64_t row0, int G, int64_t tiles,
    RowGroups rg) {
  offset = resolve_offset(offset);
  const int rpb = kBlock / G;
  const bool nts = nt_stores((int64_t)B * L * (int64_t)sizeof(T));
  const int sub = threadIdx.x % G;
  const int rloc = threadIdx.x / G;
  const int nvec = L / V;
  float acc[2] = {0.f, 0.f};  // sum mu, sum s over this thread's elements

  for (int64_t t = blockIdx.x; t < tiles; t += gridDim.x) {
    const int64_t row = t * rpb + rloc;
    float klr = 0.f;
    if (row < B) {
      const T* lrow = logits + rg.offset(row, row_stride);
      const int64_t s_off = rg.group ? rg.s_off : L;
      for (int j0 = sub; j0 < nvec; j0 += G * U) {
        float mu[U][V], s[U][V], e[U][V];
        load_chunk<T, V, U, kNtGaussFwd>(lrow, j0, G, nvec, mu);
        load_chunk<T, V, U, kNtGaussFwd>(lrow + s_off, j0, G, nvec, s);
        if (training && noise) load_chunk<T, V, U, kNtGaussFwd>(noise + row * L, j0, G, nvec, e);
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int j = j0 + u * G;
          if (j < nvec) {
            if (training && !noise)
              philox_normal<V>(seed, offset, (uint64_t)(row0 + row) * (uint64_t)L + (uint64_t)j * V, e[u]);
            float zz[V];
#pragma unroll
            for (int i = 0; i < V; ++i) {
              // logvar = raw + eps_tol, rounded to the tensor dtype as the reference's add does
              const float si = to_f32(from_f32<T>(add_rn(s[u][i], eps_tol)));
              s[u][i] = si;
              const float m = mu[u][i];
              // z = fl(fl(e * fl(0.5*s)) + mu): separate roundings, no fma, as eps.mul(std).add_(mu)
              zz[i] = training
                          ? add_rn(to_f32(from_f32<T>(mul_rn(e[u][i], to_f32(from_f32<T>(mul_rn(0.5f, si)))))), m)
                          : m;
              const float s2 = si * si;
              // log(s^2) = 2 ln2 log2|s|: |s| = |raw + eps_tol| is 0 or >= 1e-13, so bare v_log_f32 is safe
              klr += 0.5f * (s2 + m * m - 1.0f) - kLn2 * log2_raw(fabsf(si));
              acc[0] += m;
              acc[1] += si;
            }
            if (z) store_vec<T, V>(nts, z + row * z_stride + (int64_t)j * V, zz);
            if (s_out) store_vec<T, V>(nts, s_out + row * L + (int64_t)j * V, s[u]);
          }
        }
      }
    }
    klr = group_sum(klr, G);
    if (row < B && sub == 0) kl[row] = from_f32<T>(klr);
  }
  if (ws) grid_reduce<2>(acc, ws, mean_scale, means, kNanGauss2);
}

// ---------------------------------------------------------------------------
// backward: d logits from d z, d logvar, d kl and the two mean diagnostics
// ---------------------------------------------------------------------------
// At least four waves per SIMD for the one- and two-chunk forms: the 16-bit two-chunk instantiation otherwise takes
// 130 registers -- two over the 128 of occupancy 4 -- and runs at occupancy 3 (bf16 L=512: 70.5 -> 67.6 us with the
// bound, 12 bytes per lane of spill; five waves spill 156 bytes and halve the throughput: scripts/ab.py, r03).
template <typename T, int V, int U>
__global__ __launch_bounds__(kBlock, (U <= 2 ? 4 : 1)) void gauss_bwd_kernel(
    const T* __restrict__ logits, int64_t row_stride, const T* __restrict__ noise, const T* __restrict__ gz,
    int64_t gz_stride, const T* __restrict__ gs, const T* __restrict__ gkl, const float* __restrict__ gmeans,
    T* __restrict__ glogits, int64_t gl_stride, int64_t B, int L, float eps_tol, float mean_scale, int training, uint64_t seed, uint64_t offset, int64_t row0,
    int G, int64_t tiles, RowGroups rg) {
  offset = resolve_offset(offset);
  const int rpb = kBlock / G;
  const bool nts = nt_stores((int64_t)B * L * (int64_t)sizeof(T));
  const int sub = threadIdx.x % G;
  const int rloc = threadIdx.x / G;
  const int nvec = L / V;
  const float gm_mu = gmeans ? gmeans[0] * mean_scale : 0.f;
  const float gm_s = gmeans ? gmeans[1] * mean_scale : 0.f;
  const bool use_e = training && gz;  // d z / d s = 0.5 e only matters when z has a gradient

  for (int64_t t = blockIdx.x; t < tiles; t += gridDim.x) {
    const int64_t row = t * rpb + rloc;
    if (row >= B) continue;
    const T* lrow = logits + rg.offset(row, row_stride);
    T* grow = glogits + rg.offset(row, gl_stride);     // the gradient has the logits' layout
    const int64_t s_off = rg.group ? rg.s_off : L;
    const float gk = gkl ? to_f32(gkl[row]) : 0.f;
    for (int j0 = sub; j0 < nvec; j0 += G * U) {
      float mu[U][V], s[U][V], e[U][V], g[U][V], g2[U][V];
      load_chunk<T, V, U, kNtGaussBwd>(lrow, j0, G, nvec, mu);
      load_chunk<T, V, U, kNtGaussBwd>(lrow + s_off, j0, G, nvec, s);
      if (gz) load_chunk<T, V, U, kNtGaussBwd>(gz + row * gz_stride, j0, G, nvec, g);
      if (gs) load_chunk<T, V, U, kNtGaussBwd>(gs + row * L, j0, G, nvec, g2);
      if (use_e && noise) load_chunk<T, V, U, kNtGaussBwd>(noise + row * L, j0, G, nvec, e);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int j = j0 + u * G;
        if (j < nvec) {
          if (use_e && !noise)
            philox_normal<V>(seed, offset, (uint64_t)(row0 + row) * (uint64_t)L + (uint64_t)j * V, e[u]);
          float dmu[V], ds[V];
#pragma unroll
          for (int i = 0; i < V; ++i) {
            const float si = to_f32(from_f32<T>(add_rn(s[u][i], eps_tol)));
            const float gzi = gz ? g[u][i] : 0.f;
            dmu[i] = gzi + gk * mu[u][i] + gm_mu;
            ds[i] = (use_e ? 0.5f * e[u][i] * gzi : 0.f) + (gs ? g2[u][i] : 0.f) + gk * (si - fast_rcp(si)) + gm_s;
          }
          store_vec<T, V>(nts, grow + (int64_t)j * V, dmu);
          store_vec<T, V>(nts, grow + s_off + (int64_t)j * V, ds);
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------
// standalone KL between two Gaussians given as strided (mu, scale) views; the
// prior may be absent (N(0,1)).  Used when kl() is called on params that did not
// come from the fused forward (VRNN posterior-vs-prior, class-conditioned prior).
// ---------------------------------------------------------------------------
template <typename T, int V>
__global__ __launch_bounds__(kBlock) void gauss_kl_fwd_kernel(
    const T* __restrict__ mu_q, int64_t sq_mu, const T* __restrict__ s_q, int64_t sq_s, const T* __restrict__ mu_p,
    int64_t sp_mu, const T* __restrict__ s_p, int64_t sp_s, T* __restrict__ kl, int64_t B, int L, int G,
    int64_t tiles) {
  const int rpb = kBlock / G;
  const int sub = threadIdx.x % G;
  const int rloc = threadIdx.x / G;
  const int nvec = L / V;
  for (int64_t t = blockIdx.x; t < tiles; t += gridDim.x) {
    const int64_t row = t * rpb + rloc;
    float klr = 0.f;
    if (row < B) {
#pragma unroll 2
      for (int j = sub; j < nvec; j += G) {
        float mq[V], sq[V], mp[V], sp[V];
        load_vec<T, V>(mu_q + row * sq_mu + (int64_t)j * V, mq);
        load_vec<T, V>(s_q + row * sq_s + (int64_t)j * V, sq);
        if (mu_p) {
          load_vec<T, V>(mu_p + row * sp_mu + (int64_t)j * V, mp);
          load_vec<T, V>(s_p + row * sp_s + (int64_t)j * V, sp);
        }
#pragma unroll
        for (int i = 0; i < V; ++i) {
          if (mu_p) {
            // torch kl_normal_normal: var_ratio = (sq/sp)^2; t1 = ((mq-mp)/sp)^2
            const float isp = rcp_nr(sp[i]);
            float r = sq[i] * isp;
            float vr = r * r;
            float d = (mq[i] - mp[i]) * isp;
            klr += 0.5f * (vr + d * d - 1.0f - fast_log(vr));
          } else {
            float s2 = sq[i] * sq[i];
            klr += 0.5f * (s2 + mq[i] * mq[i] - 1.0f - fast_log(s2));
          }
        }
      }
    }
    klr = group_sum(klr, G);
    if (row < B && sub == 0) kl[row] = from_f32<T>(klr);
  }
}

// d kl / d (mu_q, s_q, mu_p, s_p), each written to a contiguous [B, L] buffer
// (prior gradients skipped when g_mu_p is null).
template <typename T, int V>
__global__ __launch_bounds__(kBlock) void gauss_kl_bwd_kernel(
    const T* __restrict__ mu_q, int64_t sq_mu, const T* __restrict__ s_q, int64_t sq_s, const T* __restrict__ mu_p,
    int64_t sp_mu, const T* __restrict__ s_p, int64_t sp_s, const T* __restrict__ gkl, T* __restrict__ g_mu_q,
    T* __restrict__ g_s_q, T* __restrict__ g_mu_p, T* __restrict__ g_s_p, int64_t B, int L, int G, int64_t tiles) {
  const int rpb = kBlock / G;
  const bool nts = nt_stores((int64_t)B * L * (int64_t)sizeof(T));
  const int sub = threadIdx.x % G;
  const int rloc = threadIdx.x / G;
  const int nvec = L / V;
  for (int64_t t = blockIdx.x; t < tiles; t += gridDim.x) {
    const int64_t row = t * rpb + rloc;
    if (row >= B) continue;
    const float gk = to_f32(gkl[row]);
#pragma unroll 2
    for (int j = sub; j < nvec; j += G) {
      float mq[V], sq[V], mp[V], sp[V], o0[V], o1[V], o2[V], o3[V];
      load_vec<T, V>(mu_q + row * sq_mu + (int64_t)j * V, mq);
      load_vec<T, V>(s_q + row * sq_s + (int64_t)j * V, sq);
      if (mu_p) {
        load_vec<T, V>(mu_p + row * sp_mu + (int64_t)j * V, mp);
        load_vec<T, V>(s_p + row * sp_s + (int64_t)j * V, sp);
      }
#pragma unroll
      for (int i = 0; i < V; ++i) {
        if (mu_p) {
          float ip = 1.0f / sp[i];
          float ip2 = ip * ip;
          float d = mq[i] - mp[i];
          // kl = 0.5*(sq^2/sp^2 + d^2/sp^2 - 1 - log(sq^2) + log(sp^2))
          o0[i] = gk * d * ip2;
          o1[i] = gk * (sq[i] * ip2 - 1.0f / sq[i]);
          o2[i] = -o0[i];
          o3[i] = gk * (ip - (sq[i] * sq[i] + d * d) * ip2 * ip);
        } else {
          o0[i] = gk * mq[i];
          o1[i] = gk * (sq[i] - 1.0f / sq[i]);
        }
      }
      const int64_t o = row * L + (int64_t)j * V;
      store_vec<T, V>(nts, g_mu_q + o, o0);
      store_vec<T, V>(nts, g_s_q + o, o1);
      if (mu_p && g_mu_p) {
        store_vec<T, V>(nts, g_mu_p + o, o2);
        store_vec<T, V>(nts, g_s_p + o, o3);
      }
    }
  }
}

// Launch geometry of the two Gaussian kernels on top of row_geom's general rule, from the r02 cold-cache sweep
// (scripts/sweep_geom.py; profiles/r02_geom_sweep.txt): rows of 65..128 vectors run two rows per wave in the forward
// (G = 32, two chunks in flight: fp32 L = 512 96.6 -> 94.1 us, bf16 L = 1024 101.9 -> 95.5 us), and a kernel that
// draws its noise from Philox keeps ONE chunk in flight (the generator's registers cost occupancy: fp32 L = 512
// forward 79.0 -> 72.6 us, backward 111.2 -> 106.1 us; rows above 128 vectors keep four chunks in the backward).
static RowGeom gauss_geom(int64_t B, int64_t vecs, int64_t max_grid, bool philox, bool fwd) {
  RowGeom g = row_geom(B, vecs, max_grid);
  if (knobs().G || knobs().U) return g;
  if (fwd && vecs > 64 && vecs <= 128) regroup(g, B, 32, 2, max_grid);
  if (philox && (fwd || vecs <= 128)) g.U = 1;
  return g;
}

template <typename T>
static int gauss_fwd_dispatch(const void* logits, int64_t row_stride, const void* noise, void* z, int64_t z_stride,
                              void* s_out,
                              void* kl, float* means, float* workspace, int64_t B, int64_t L, float eps_tol,
                              float mean_scale, int training, uint64_t seed, uint64_t offset, int64_t row0,
                              hipStream_t stream, RowGroups rg = RowGroups{0, 0, 0}) {
  constexpr int VN = VecOf<T>::n;
  const bool vec = (L % VN == 0) && (row_stride % VN == 0) && (z_stride % VN == 0) && aligned16(logits) &&
                   (!z || aligned16(z)) && (rg.group_stride % VN == 0) && (rg.s_off % VN == 0) &&
                   (!noise || aligned16(noise)) && (!s_out || aligned16(s_out));
  const RowGeom g = gauss_geom(B, vec ? L / VN : L, means ? kGridReduce : kGridStream, training && !noise, true);
  if (!means) workspace = nullptr;
  if (vec) {
    VAE_SWITCH_U(g.U, hipLaunchKernelGGL((gauss_fwd_kernel<T, VN, U>), dim3(g.grid), dim3(kBlock), 0, stream,
                                         (const T*)logits, row_stride, (const T*)noise, (T*)z, z_stride, (T*)s_out, (T*)kl,
                                         workspace, mean_scale, means, B, (int)L, eps_tol, training, seed, offset,
                                         row0, g.G, g.tiles, rg));
  } else {
    VAE_SWITCH_U(g.U, hipLaunchKernelGGL((gauss_fwd_kernel<T, 1, U>), dim3(g.grid), dim3(kBlock), 0, stream,
                                         (const T*)logits, row_stride, (const T*)noise, (T*)z, z_stride, (T*)s_out, (T*)kl,
                                         workspace, mean_scale, means, B, (int)L, eps_tol, training, seed, offset,
                                         row0, g.G, g.tiles, rg));
  }
  return hipGetLastError() == hipSuccess ? VAE_OK : VAE_ERR_LAUNCH;
}

template <typename T>
static int gauss_bwd_dispatch(const void* logits, int64_t row_stride, const void* noise, const void* gz,
                              int64_t gz_stride, const void* gs, const void* gkl, const float* gmeans, void* glogits,
                              int64_t gl_stride, int64_t B,
                              int64_t L, float eps_tol, float mean_scale, int training, uint64_t seed,
                              uint64_t offset, int64_t row0, hipStream_t stream, RowGroups rg = RowGroups{0, 0, 0}) {
  constexpr int VN = VecOf<T>::n;
  const bool vec = (L % VN == 0) && (row_stride % VN == 0) && (gz_stride % VN == 0) && (gl_stride % VN == 0) &&
                   aligned16(logits) && aligned16(glogits) && (rg.group_stride % VN == 0) && (rg.s_off % VN == 0) &&
                   (!noise || aligned16(noise)) && (!gz || aligned16(gz)) && (!gs || aligned16(gs));
  const RowGeom g = gauss_geom(B, vec ? L / VN : L, kGridStream, training && !noise, false);
  if (vec) {
    VAE_SWITCH_U(g.U, hipLaunchKernelGGL((gauss_bwd_kernel<T, VN, U>), dim3(g.grid), dim3(kBlock), 0, stream,
                                         (const T*)logits, row_stride, (const T*)noise, (const T*)gz, gz_stride,
                                         (const T*)gs, (const T*)gkl, gmeans, (T*)glogits, gl_stride, B, (int)L, eps_tol, mean_scale, training,
                                         seed, offset, row0, g.G, g.tiles, rg));
  } else {
    VAE_SWITCH_U(g.U, hipLaunchKernelGGL((gauss_bwd_kernel<T, 1, U>), dim3(g.grid), dim3(kBlock), 0, stream,
                                         (const T*)logits, row_stride, (const T*)noise, (const T*)gz, gz_stride,
                                         (const T*)gs, (const T*)gkl, gmeans, (T*)glogits, gl_stride, B, (int)L, eps_tol, mean_scale, training,
                                         seed, offset, row0, g.G, g.tiles, rg));
  }
  return hipGetLastError() == hipSuccess ? VAE_OK : VAE_ERR_LAUNCH;
}

template <typename T>
static int gauss_kl_fwd_dispatch(const void* mu_q, int64_t sq_mu, const void* s_q, int64_t sq_s, const void* mu_p,
                                 int64_t sp_mu, const void* s_p, int64_t sp_s, void* kl, int64_t B, int64_t L,
                                 hipStream_t stream) {
  constexpr int VN = VecOf<T>::n;
  bool vec = (L % VN == 0) && (sq_mu % VN == 0) && (sq_s % VN == 0) && aligned16(mu_q) && aligned16(s_q);
  if (mu_p) vec = vec && (sp_mu % VN == 0) && (sp_s % VN == 0) && aligned16(mu_p) && aligned16(s_p);
  const RowGeom g = row_geom(B, vec ? L / VN : L);
  if (vec) {
    hipLaunchKernelGGL((gauss_kl_fwd_kernel<T, VN>), dim3(g.grid), dim3(kBlock), 0, stream, (const T*)mu_q, sq_mu,
                       (const T*)s_q, sq_s, (const T*)mu_p, sp_mu, (const T*)s_p, sp_s, (T*)kl, B, (int)L, g.G,
                       g.tiles);
  } else {
    hipLaunchKernelGGL((gauss_kl_fwd_kernel<T, 1>), dim3(g.grid), dim3(kBlock), 0, stream, (const T*)mu_q, sq_mu,
                       (const T*)s_q, sq_s, (const T*)mu_p, sp_mu, (const T*)s_p, sp_s, (T*)kl, B, (int)L, g.G,
                       g.tiles);
  }
  return hipGetLastError() == hipSuccess ? VAE_OK : VAE_ERR_LAUNCH;
}

template <typename T>
static int gauss_kl_bwd_dispatch(const void* mu_q, int64_t sq_mu, const void* s_q, int64_t sq_s, const void* mu_p,
                                 int64_t sp_mu, const void* s_p, int64_t sp_s, const void* gkl, void* g_mu_q,
                                 void* g_s_q, void* g_mu_p, void* g_s_p, int64_t B, int64_t L, hipStream_t stream) {
  constexpr int VN = VecOf<T>::n;
  bool vec = (L % VN == 0) && (sq_mu % VN == 0) && (sq_s % VN == 0) && aligned16(mu_q) && aligned16(s_q) &&
             aligned16(g_mu_q) && aligned16(g_s_q);
  if (mu_p) vec = vec && (sp_mu % VN == 0) && (sp_s % VN == 0) && aligned16(mu_p) && aligned16(s_p);
  if (g_mu_p) vec = vec && aligned16(g_mu_p) && aligned16(g_s_p);
  const RowGeom g = row_geom(B, vec ? L / VN : L);
  if (vec) {
    hipLaunchKernelGGL((gauss_kl_bwd_kernel<T, VN>), dim3(g.grid), dim3(kBlock), 0, stream, (const T*)mu_q, sq_mu,
                       (const T*)s_q, sq_s, (const T*)mu_p, sp_mu, (const T*)s_p, sp_s, (const T*)gkl, (T*)g_mu_q,
                       (T*)g_s_q, (T*)g_mu_p, (T*)g_s_p, B, (int)L, g.G, g.tiles);
  } else {
    hipLaunchKernelGGL((gauss_kl_bwd_kernel<T, 1>), dim3(g.grid), dim3(kBlock), 0, stream, (const T*)mu_q, sq_mu,
                       (const T*)s_q, sq_s, (const T*)mu_p, sp_mu, (const T*)s_p, sp_s, (const T*)gkl, (T*)g_mu_q,
                       (T*)g_s_q, (T*)g_mu_p, (T*)g_s_p, B, (int)L, g.G, g.tiles);
  }
  return hipGetLastError() == hipSuccess ? VAE_OK : VAE_ERR_LAUNCH;
}

}  // namespace vae

using namespace vae;

extern "C" {

int vae_gauss_reparam_kl_fwd(const void* logits, int64_t row_stride, const void* noise, void* z, int64_t z_stride,
                             void* s_out,
                             void* kl, float* means, float* workspace, int64_t B, int64_t L, float eps_tol,
                             float mean_scale, int training, uint64_t seed, uint64_t offset, int64_t row0, int dtype,
                             vae_stream_t stream) {
  if (B < 0 || L <= 0 || L > VAE_MAX_ROW) return VAE_ERR_ARG;
  if (B == 0) return VAE_OK;
  if (z_stride == 0) z_stride = L;
  if (!logits || (!z && training) || !kl || !workspace || row_stride < 2 * L || z_stride < L) return VAE_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == VAE_F32)
    return gauss_fwd_dispatch<float>(logits, row_stride, noise, z, z_stride, s_out, kl, means, workspace, B, L, eps_tol,
                                     mean_scale, training, seed, offset, row0, st);
  if (dtype == VAE_BF16)
    return gauss_fwd_dispatch<bf16_t>(logits, row_stride, noise, z, z_stride, s_out, kl, means, workspace, B, L, eps_tol,
                                      mean_scale, training, seed, offset, row0, st);
  if (dtype == VAE_F16)
    return gauss_fwd_dispatch<f16_t>(logits, row_stride, noise, z, z_stride, s_out, kl, means, workspace, B, L, eps_tol,
                                      mean_scale, training, seed, offset, row0, st);
  return VAE_ERR_ARG;
}

int vae_gauss_reparam_kl_bwd(const void* logits, int64_t row_stride, const void* noise, const void* gz,
                             int64_t gz_stride, const void* gs, const void* gkl, const float* gmeans, void* glogits,
                             int64_t glogits_stride, int64_t B,
                             int64_t L, float eps_tol, float mean_scale, int training, uint64_t seed,
                             uint64_t offset, int64_t row0, int dtype, vae_stream_t stream) {
  if (B < 0 || L <= 0 || L > VAE_MAX_ROW) return VAE_ERR_ARG;
  if (B == 0) return VAE_OK;
  if (gz_stride == 0) gz_stride = L;
  if (glogits_stride == 0) glogits_stride = 2 * L;
  if (!logits || !glogits || row_stride < 2 * L || gz_stride < L || glogits_stride < 2 * L) return VAE_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == VAE_F32)
    return gauss_bwd_dispatch<float>(logits, row_stride, noise, gz, gz_stride, gs, gkl, gmeans, glogits, glogits_stride, B, L, eps_tol,
                                     mean_scale, training, seed, offset, row0, st);
  if (dtype == VAE_BF16)
    return gauss_bwd_dispatch<bf16_t>(logits, row_stride, noise, gz, gz_stride, gs, gkl, gmeans, glogits, glogits_stride,
                                      B, L, eps_tol,
                                      mean_scale, training, seed, offset, row0, st);
  if (dtype == VAE_F16)
    return gauss_bwd_dispatch<f16_t>(logits, row_stride, noise, gz, gz_stride, gs, gkl, gmeans, glogits, glogits_stride,
                                      B, L, eps_tol,
                                      mean_scale, training, seed, offset, row0, st);
  return VAE_ERR_ARG;
}

int vae_gauss_reparam_kl_fwd_nchw(const void* logits, const void* noise, void* z, void* s_out, void* kl, float* means,
                                  float* workspace, int64_t B, int64_t C, int64_t H, int64_t W, float eps_tol,
                                  float mean_scale, int training, uint64_t seed, uint64_t offset, int64_t row0,
                                  int dtype, vae_stream_t stream) {
  if (B < 0 || C <= 0 || H <= 0 || W <= 0 || W > VAE_MAX_ROW) return VAE_ERR_ARG;
  if (B == 0) return VAE_OK;
  if (!logits || (!z && training) || !kl || !workspace) return VAE_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  const RowGroups rg{C * H, 2 * C * H * W, C * H * W};
  const int64_t rows = B * C * H;
  if (dtype == VAE_F32)
    return gauss_fwd_dispatch<float>(logits, W, noise, z, W, s_out, kl, means, workspace, rows, W, eps_tol, mean_scale,
                                     training, seed, offset, row0, st, rg);
  if (dtype == VAE_BF16)
    return gauss_fwd_dispatch<bf16_t>(logits, W, noise, z, W, s_out, kl, means, workspace, rows, W, eps_tol, mean_scale,
                                      training, seed, offset, row0, st, rg);
  if (dtype == VAE_F16)
    return gauss_fwd_dispatch<f16_t>(logits, W, noise, z, W, s_out, kl, means, workspace, rows, W, eps_tol, mean_scale,
                                     training, seed, offset, row0, st, rg);
  return VAE_ERR_ARG;
}

int vae_gauss_reparam_kl_bwd_nchw(const void* logits, const void* noise, const void* gz, const void* gs, const void* gkl,
                                  const float* gmeans, void* glogits, int64_t B, int64_t C, int64_t H, int64_t W,
                                  float eps_tol, float mean_scale, int training, uint64_t seed, uint64_t offset,
                                  int64_t row0, int dtype, vae_stream_t stream) {
  if (B < 0 || C <= 0 || H <= 0 || W <= 0 || W > VAE_MAX_ROW) return VAE_ERR_ARG;
  if (B == 0) return VAE_OK;
  if (!logits || !glogits) return VAE_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  const RowGroups rg{C * H, 2 * C * H * W, C * H * W};
  const int64_t rows = B * C * H;
  if (dtype == VAE_F32)
    return gauss_bwd_dispatch<float>(logits, W, noise, gz, W, gs, gkl, gmeans, glogits, W, rows, W, eps_tol, mean_scale,
                                     training, seed, offset, row0, st, rg);
  if (dtype == VAE_BF16)
    return gauss_bwd_dispatch<bf16_t>(logits, W, noise, gz, W, gs, gkl, gmeans, glogits, W, rows, W, eps_tol, mean_scale,
                                      training, seed, offset, row0, st, rg);
  if (dtype == VAE_F16)
    return gauss_bwd_dispatch<f16_t>(logits, W, noise, gz, W, gs, gkl, gmeans, glogits, W, rows, W, eps_tol, mean_scale,
                                     training, seed, offset, row0, st, rg);
  return VAE_ERR_ARG;
}

int vae_gauss_kl_fwd(const void* mu_q, int64_t stride_mu_q, const void* s_q, int64_t stride_s_q, const void* mu_p,
                     int64_t stride_mu_p, const void* s_p, int64_t stride_s_p, void* kl, int64_t B, int64_t L,
                     int dtype, vae_stream_t stream) {
  if (B < 0 || L <= 0 || L > VAE_MAX_ROW) return VAE_ERR_ARG;
  if (B == 0) return VAE_OK;
  if (!mu_q || !s_q || !kl || ((mu_p == nullptr) != (s_p == nullptr))) return VAE_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == VAE_F32)
    return gauss_kl_fwd_dispatch<float>(mu_q, stride_mu_q, s_q, stride_s_q, mu_p, stride_mu_p, s_p, stride_s_p, kl,
                                        B, L, st);
  if (dtype == VAE_BF16)
    return gauss_kl_fwd_dispatch<bf16_t>(mu_q, stride_mu_q, s_q, stride_s_q, mu_p, stride_mu_p, s_p, stride_s_p, kl,
                                         B, L, st);
  if (dtype == VAE_F16)
    return gauss_kl_fwd_dispatch<f16_t>(mu_q, stride_mu_q, s_q, stride_s_q, mu_p, stride_mu_p, s_p, stride_s_p, kl,
                                         B, L, st);
  return VAE_ERR_ARG;
}

int vae_gauss_kl_bwd(const void* mu_q, int64_t stride_mu_q, const void* s_q, int64_t stride_s_q, const void* mu_p,
                     int64_t stride_mu_p, const void* s_p, int64_t stride_s_p, const void* gkl, void* g_mu_q,
                     void* g_s_q, void* g_mu_p, void* g_s_p, int64_t B, int64_t L, int dtype, vae_stream_t stream) {
  if (B < 0 || L <= 0 || L > VAE_MAX_ROW) return VAE_ERR_ARG;
  if (B == 0) return VAE_OK;
  if (!mu_q || !s_q || !gkl || !g_mu_q || !g_s_q || ((mu_p == nullptr) != (s_p == nullptr)) ||
      ((g_mu_p == nullptr) != (g_s_p == nullptr)) || (g_mu_p && !mu_p))
    return VAE_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == VAE_F32)
    return gauss_kl_bwd_dispatch<float>(mu_q, stride_mu_q, s_q, stride_s_q, mu_p, stride_mu_p, s_p, stride_s_p, gkl,
                                        g_mu_q, g_s_q, g_mu_p, g_s_p, B, L, st);
  if (dtype == VAE_BF16)
    return gauss_kl_bwd_dispatch<bf16_t>(mu_q, stride_mu_q, s_q, stride_s_q, mu_p, stride_mu_p, s_p, stride_s_p,
                                         gkl, g_mu_q, g_s_q, g_mu_p, g_s_p, B, L, st);
  if (dtype == VAE_F16)
    return gauss_kl_bwd_dispatch<f16_t>(mu_q, stride_mu_q, s_q, stride_s_q, mu_p, stride_mu_p, s_p, stride_s_p,
                                         gkl, g_mu_q, g_s_q, g_mu_p, g_s_p, B, L, st);
  return VAE_ERR_ARG;
}

}  // extern "C"
