// Multi-sample ("same logits, many noises") forms of the Gaussian reparameterizer and the NLL+ELBO.
//
// Replaces, from /root/reference:
//   abstract_vae.py:347-354   loss_function(K > 1): K-1 further reparameterize(logits) + decode + nll
//                             calls, nll = (nll_0 + ... + nll_{K-1}) / K
//   msg.py:55-69              MSGVAE.reparameterize: max_time_steps draws from the same logits
//   abstract_vae.py:301-311   likelihood(): repeated draws from one expanded logits row
// The looped reference re-reads the logits (8L) and rewrites logvar (4L) and the KL per draw, and re-reads x
// (4D) per NLL; here S draws cost one logits read + S (noise, z) streams, and the S-sample NLL reads x once.
// Per sample the forward moves 8L + 4L + S*(4L + 4L) + 4 and 4D + S*4D(8D) + 12 bytes.
#include "common.hpp"
#include "nll_math.hpp"
#include "../../include/vae_amd.h"

namespace vae {

// ---------------------------------------------------------------------------
// S Gaussian draws: z[s] = e[s] * fl(0.5 * s) + mu, KL and means once.  noise / z are [S, B, L].
// Philox draw s uses call offset (offset + s): the same stream S consecutive single calls would use.
// ---------------------------------------------------------------------------
template <typename T, int V, int U>
__global__ __launch_bounds__(kBlock) void gauss_multi_fwd_kernel(
    const T* __restrict__ logits, int64_t row_stride, const T* __restrict__ noise, T* __restrict__ z,
    T* __restrict__ s_out, T* __restrict__ kl, float* __restrict__ ws, float mean_scale, float* __restrict__ means,
    int64_t B, int L, int S, float eps_tol, uint64_t seed, uint64_t offset, int64_t row0, int G, int64_t tiles) {
  offset = resolve_offset(offset);
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
        float mu[U][V], s[U][V], hs[U][V];
        load_chunk<T, V, U>(lrow, j0, G, nvec, mu);
        load_chunk<T, V, U>(lrow + L, j0, G, nvec, s);
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int j = j0 + u * G;
          if (j < nvec) {
#pragma unroll
            for (int i = 0; i < V; ++i) {
              const float si = to_f32(from_f32<T>(add_rn(s[u][i], eps_tol)));
              s[u][i] = si;
              hs[u][i] = to_f32(from_f32<T>(mul_rn(0.5f, si)));
              const float m = mu[u][i];
              klr += 0.5f * (si * si + m * m - 1.0f) - kLn2 * log2_raw(fabsf(si));
              acc[0] += m;
              acc[1] += si;
            }
            if (s_out) store_vec<T, V>(nts, s_out + row * L + (int64_t)j * V, s[u]);
          }
        }
        for (int smp = 0; smp < S; ++smp) {
          const int64_t so = ((int64_t)smp * B + row) * L;
          float e[U][V];
          if (noise) load_chunk<T, V, U>(noise + so, j0, G, nvec, e);
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const int j = j0 + u * G;
            if (j < nvec) {
              if (!noise)
                philox_normal<V>(seed, offset + (uint64_t)smp, (uint64_t)(row0 + row) * (uint64_t)L + (uint64_t)j * V, e[u]);
              float zz[V];
#pragma unroll
              for (int i = 0; i < V; ++i) zz[i] = add_rn(to_f32(from_f32<T>(mul_rn(e[u][i], hs[u][i]))), mu[u][i]);
              store_vec<T, V>(nts, z + so + (int64_t)j * V, zz);
            }
          }
        }
      }
    }
    klr = group_sum(klr, G);
    if (row < B && sub == 0) kl[row] = from_f32<T>(klr);
  }
  if (ws) grid_reduce<2>(acc, ws, mean_scale, means, kNanGauss2);
}

// d mu = sum_s gz[s] + gkl*mu + gm;  d s = 0.5 sum_s e[s] gz[s] + gs + gkl (s - 1/s) + gm
template <typename T, int V, int U>
__global__ __launch_bounds__(kBlock) void gauss_multi_bwd_kernel(
    const T* __restrict__ logits, int64_t row_stride, const T* __restrict__ noise, const T* __restrict__ gz,
    const T* __restrict__ gs, const T* __restrict__ gkl, const float* __restrict__ gmeans, T* __restrict__ glogits,
    int64_t B, int L, int S, float eps_tol, float mean_scale, uint64_t seed, uint64_t offset, int64_t row0, int G,
    int64_t tiles) {
  offset = resolve_offset(offset);
  const int rpb = kBlock / G;
  const bool nts = nt_stores((int64_t)B * L * (int64_t)sizeof(T));
  const int sub = threadIdx.x % G;
  const int rloc = threadIdx.x / G;
  const int nvec = L / V;
  const float gm_mu = gmeans ? gmeans[0] * mean_scale : 0.f;
  const float gm_s = gmeans ? gmeans[1] * mean_scale : 0.f;
  for (int64_t t = blockIdx.x; t < tiles; t += gridDim.x) {
    const int64_t row = t * rpb + rloc;
    if (row >= B) continue;
    const T* lrow = logits + row * row_stride;
    T* grow = glogits + row * 2 * (int64_t)L;
    const float gk = gkl ? to_f32(gkl[row]) : 0.f;
    for (int j0 = sub; j0 < nvec; j0 += G * U) {
      float mu[U][V], s[U][V], g2[U][V], am[U][V], as[U][V];
      load_chunk<T, V, U>(lrow, j0, G, nvec, mu);
      load_chunk<T, V, U>(lrow + L, j0, G, nvec, s);
      if (gs) load_chunk<T, V, U>(gs + row * L, j0, G, nvec, g2);
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int i = 0; i < V; ++i) am[u][i] = as[u][i] = 0.f;
      if (gz) {
        for (int smp = 0; smp < S; ++smp) {
          const int64_t so = ((int64_t)smp * B + row) * L;
          float g[U][V], e[U][V];
          load_chunk<T, V, U>(gz + so, j0, G, nvec, g);
          if (noise) load_chunk<T, V, U>(noise + so, j0, G, nvec, e);
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const int j = j0 + u * G;
            if (j < nvec) {
              if (!noise)
                philox_normal<V>(seed, offset + (uint64_t)smp, (uint64_t)(row0 + row) * (uint64_t)L + (uint64_t)j * V, e[u]);
#pragma unroll
              for (int i = 0; i < V; ++i) {
                am[u][i] += g[u][i];
                as[u][i] += e[u][i] * g[u][i];
              }
            }
          }
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int j = j0 + u * G;
        if (j < nvec) {
          float dmu[V], ds[V];
#pragma unroll
          for (int i = 0; i < V; ++i) {
            const float si = to_f32(from_f32<T>(add_rn(s[u][i], eps_tol)));
            dmu[i] = am[u][i] + gk * mu[u][i] + gm_mu;
            ds[i] = 0.5f * as[u][i] + (gs ? g2[u][i] : 0.f) + gk * (si - fast_rcp(si)) + gm_s;
          }
          store_vec<T, V>(nts, grow + (int64_t)j * V, dmu);
          store_vec<T, V>(nts, grow + L + (int64_t)j * V, ds);
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------
// S-sample NLL + ELBO: nll[b] = (1/S) sum_s nll(x[b], recon_s[b]); x is read once per row chunk.
// The S decoder outputs are separate tensors (one decode call each, as in the reference), handed over
// as a by-value pointer pack; the pick below is a uniform select chain (no private-memory indexing).
// ---------------------------------------------------------------------------
struct PtrPack {
  void* p[VAE_MAX_SAMPLES];
};
__device__ __forceinline__ void* pick(const PtrPack& k, int s) {
  void* r = k.p[0];
#pragma unroll
  for (int i = 1; i < VAE_MAX_SAMPLES; ++i) r = (s == i) ? k.p[i] : r;
  return r;
}

template <typename T, int V, int NLL, int U>
__global__ __launch_bounds__(kBlock) void nll_multi_fwd_kernel(
    const T* __restrict__ x, PtrPack recons, int S, const T* __restrict__ kld, const T* __restrict__ add,
    const T* __restrict__ mi, T* __restrict__ nll_out, T* __restrict__ loss, T* __restrict__ elbo, int64_t B, int D,
    float kl_beta, int G, int64_t tiles) {
  const int rpb = kBlock / G;
  const int sub = threadIdx.x % G;
  const int rloc = threadIdx.x / G;
  const int nvec = D / V;
  const int64_t rstride = (NLL == VAE_NLL_BERNOULLI) ? (int64_t)D : 2 * (int64_t)D;
  const float inv_s = 1.0f / (float)S;
  for (int64_t t = blockIdx.x; t < tiles; t += gridDim.x) {
    const int64_t row = t * rpb + rloc;
    float s = 0.f;
    float k = 0.f, ad = 0.f, m = 0.f;   // requested with the row (see nll_elbo_fused_kernel)
    if (row < B) {
      if (loss) {
        if (kld) k = to_f32(kld[row]);
        if (add) ad = to_f32(add[row]);
        if (mi) m = to_f32(mi[row]);
      }
      const T* xr = x + row * D;
      for (int j0 = sub; j0 < nvec; j0 += G * U) {
        float xv[U][V];
        load_chunk<T, V, U>(xr, j0, G, nvec, xv);
        for (int smp = 0; smp < S; ++smp) {
          const T* rr = (const T*)pick(recons, smp) + row * rstride;
          float a[U][V], b[U][V];
          load_chunk<T, V, U>(rr, j0, G, nvec, a);
          if constexpr (NLL != VAE_NLL_BERNOULLI) load_chunk<T, V, U>(rr + D, j0, G, nvec, b);
#pragma unroll
          for (int u = 0; u < U; ++u) {
            float su = 0.f;
#pragma unroll
            for (int i = 0; i < V; ++i) {
              if constexpr (NLL == VAE_NLL_BERNOULLI) su += nll_bern(xv[u][i], a[u][i]);
              if constexpr (NLL == VAE_NLL_GAUSSIAN) su += nll_gauss(xv[u][i], a[u][i], b[u][i]);
              if constexpr (NLL == VAE_NLL_DISC_LOGISTIC) su += nll_dlog(xv[u][i], a[u][i], b[u][i]);
            }
            s += (j0 + u * G < nvec) ? su : 0.f;
          }
        }
      }
    }
    s = row_sum(s, G);
    if (row < B && sub == 0) {
      const float n = to_f32(from_f32<T>(s * inv_s));
      nll_out[row] = from_f32<T>(n);
      if (loss) {
        const float e = to_f32(from_f32<T>(n + k));
        const float l = to_f32(from_f32<T>((n + ad + kl_beta * k) - m));
        loss[row] = from_f32<T>(l);
        elbo[row] = from_f32<T>(e);
      }
    }
  }
}

template <typename T, int V, int NLL, int U>
__global__ __launch_bounds__(kBlock) void nll_multi_bwd_kernel(
    const T* __restrict__ x, PtrPack recons, PtrPack grecons, int S, const T* __restrict__ gnll,
    const T* __restrict__ gloss, const T* __restrict__ gelbo, const float* __restrict__ gmeans, T* __restrict__ gkld,
    T* __restrict__ gadd, T* __restrict__ gmi, int64_t B, int D, float kl_beta, float inv_count, int G,
    int64_t tiles) {
  const int rpb = kBlock / G;
  const bool nts = nt_stores((int64_t)B * D * (int64_t)sizeof(T));
  const int sub = threadIdx.x % G;
  const int rloc = threadIdx.x / G;
  const int nvec = D / V;
  const int64_t rstride = (NLL == VAE_NLL_BERNOULLI) ? (int64_t)D : 2 * (int64_t)D;
  const float inv_s = 1.0f / (float)S;
  float gm[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) gm[k] = gmeans ? gmeans[k] * inv_count : 0.f;
  for (int64_t t = blockIdx.x; t < tiles; t += gridDim.x) {
    const int64_t row = t * rpb + rloc;
    if (row >= B) continue;
    const float gl = (gloss ? to_f32(gloss[row]) : 0.f) + gm[0];
    const float ge = (gelbo ? to_f32(gelbo[row]) : 0.f) + gm[1];
    const float c = ((gnll ? to_f32(gnll[row]) : 0.f) + gl + ge + gm[2]) * inv_s;
    if (sub == 0) {
      if (gkld) gkld[row] = from_f32<T>(kl_beta * gl + ge + gm[3]);
      if (gadd) gadd[row] = from_f32<T>(gl + gm[4]);
      if (gmi) gmi[row] = from_f32<T>(gm[5] - gl);
    }
    const T* xr = x + row * D;
    for (int j0 = sub; j0 < nvec; j0 += G * U) {
      float xv[U][V];
      load_chunk<T, V, U>(xr, j0, G, nvec, xv);
      for (int smp = 0; smp < S; ++smp) {
        const T* rr = (const T*)pick(recons, smp) + row * rstride;
        T* gr = (T*)pick(grecons, smp) + row * rstride;
        float a[U][V], b[U][V];
        load_chunk<T, V, U>(rr, j0, G, nvec, a);
        if constexpr (NLL != VAE_NLL_BERNOULLI) load_chunk<T, V, U>(rr + D, j0, G, nvec, b);
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int j = j0 + u * G;
          if (j < nvec) {
            float ga[V], gb[V];
#pragma unroll
            for (int i = 0; i < V; ++i) {
              if constexpr (NLL == VAE_NLL_BERNOULLI) ga[i] = c * nll_bern_grad(xv[u][i], a[u][i]);
              if constexpr (NLL == VAE_NLL_GAUSSIAN) {
                nll_gauss_grad(xv[u][i], a[u][i], b[u][i], ga[i], gb[i]);
                ga[i] *= c; gb[i] *= c;
              }
              if constexpr (NLL == VAE_NLL_DISC_LOGISTIC) {
                nll_dlog_grad(xv[u][i], a[u][i], b[u][i], ga[i], gb[i]);
                ga[i] *= c; gb[i] *= c;
              }
            }
            store_vec<T, V>(nts, gr + (int64_t)j * V, ga);
            if constexpr (NLL != VAE_NLL_BERNOULLI) store_vec<T, V>(nts, gr + D + (int64_t)j * V, gb);
          }
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
template <typename T>
static int gauss_multi_fwd_dispatch(const void* logits, int64_t rs, const void* noise, void* z, void* s_out, void* kl,
                                    float* means, float* ws, int64_t B, int64_t L, int S, float eps_tol,
                                    float mean_scale, uint64_t seed, uint64_t offset, int64_t row0, hipStream_t st) {
  constexpr int VN = VecOf<T>::n;
  const bool vec = (L % VN == 0) && (rs % VN == 0) && aligned16(logits) && aligned16(z) &&
                   (!noise || aligned16(noise)) && (!s_out || aligned16(s_out));
  const RowGeom g = row_geom(B, vec ? L / VN : L, kGridReduce);
  float* partials = means ? ws : nullptr;
  if (vec) {
    VAE_SWITCH_U(g.U, hipLaunchKernelGGL((gauss_multi_fwd_kernel<T, VN, U>), dim3(g.grid), dim3(kBlock), 0, st,
                                         (const T*)logits, rs, (const T*)noise, (T*)z, (T*)s_out, (T*)kl, partials,
                                         mean_scale, means, B, (int)L, S, eps_tol, seed, offset, row0, g.G, g.tiles));
  } else {
    VAE_SWITCH_U(g.U, hipLaunchKernelGGL((gauss_multi_fwd_kernel<T, 1, U>), dim3(g.grid), dim3(kBlock), 0, st,
                                         (const T*)logits, rs, (const T*)noise, (T*)z, (T*)s_out, (T*)kl, partials,
                                         mean_scale, means, B, (int)L, S, eps_tol, seed, offset, row0, g.G, g.tiles));
  }
  return hipGetLastError() == hipSuccess ? VAE_OK : VAE_ERR_LAUNCH;
}

template <typename T>
static int gauss_multi_bwd_dispatch(const void* logits, int64_t rs, const void* noise, const void* gz, const void* gs,
                                    const void* gkl, const float* gmeans, void* glogits, int64_t B, int64_t L, int S,
                                    float eps_tol, float mean_scale, uint64_t seed, uint64_t offset, int64_t row0,
                                    hipStream_t st) {
  constexpr int VN = VecOf<T>::n;
  const bool vec = (L % VN == 0) && (rs % VN == 0) && aligned16(logits) && aligned16(glogits) &&
                   (!noise || aligned16(noise)) && (!gz || aligned16(gz)) && (!gs || aligned16(gs));
  const RowGeom g = row_geom(B, vec ? L / VN : L, kGridStream);
  if (vec) {
    VAE_SWITCH_U(g.U, hipLaunchKernelGGL((gauss_multi_bwd_kernel<T, VN, U>), dim3(g.grid), dim3(kBlock), 0, st,
                                         (const T*)logits, rs, (const T*)noise, (const T*)gz, (const T*)gs,
                                         (const T*)gkl, gmeans, (T*)glogits, B, (int)L, S, eps_tol, mean_scale, seed,
                                         offset, row0, g.G, g.tiles));
  } else {
    VAE_SWITCH_U(g.U, hipLaunchKernelGGL((gauss_multi_bwd_kernel<T, 1, U>), dim3(g.grid), dim3(kBlock), 0, st,
                                         (const T*)logits, rs, (const T*)noise, (const T*)gz, (const T*)gs,
                                         (const T*)gkl, gmeans, (T*)glogits, B, (int)L, S, eps_tol, mean_scale, seed,
                                         offset, row0, g.G, g.tiles));
  }
  return hipGetLastError() == hipSuccess ? VAE_OK : VAE_ERR_LAUNCH;
}

static bool pack_ptrs(const void* const* src, int S, PtrPack* out, bool* all_aligned) {
  for (int i = 0; i < VAE_MAX_SAMPLES; ++i) out->p[i] = nullptr;
  for (int i = 0; i < S; ++i) {
    if (!src[i]) return false;
    out->p[i] = const_cast<void*>(src[i]);
    if (!aligned16(src[i])) *all_aligned = false;
  }
  return true;
}

template <typename T, int NLL>
static int nll_multi_fwd_dispatch(const void* x, const PtrPack& rp, bool al, int S, const void* kld, const void* add,
                                  const void* mi, void* nll, void* loss, void* elbo, float* means, float* ws,
                                  int64_t B, int64_t D, float kl_beta, float inv_count, hipStream_t st) {
  constexpr int VN = VecOf<T>::n;
  const bool vec = (D % VN == 0) && aligned16(x) && al;
  const RowGeom g = nll_geom(B, vec ? D / VN : D, kGridStream, false, true);
  if (vec) {
    VAE_SWITCH_U(g.U, hipLaunchKernelGGL((nll_multi_fwd_kernel<T, VN, NLL, U>), dim3(g.grid), dim3(kBlock), 0, st,
                                         (const T*)x, rp, S, (const T*)kld, (const T*)add, (const T*)mi, (T*)nll,
                                         (T*)loss, (T*)elbo, B, (int)D, kl_beta, g.G, g.tiles));
  } else {
    VAE_SWITCH_U(g.U, hipLaunchKernelGGL((nll_multi_fwd_kernel<T, 1, NLL, U>), dim3(g.grid), dim3(kBlock), 0, st,
                                         (const T*)x, rp, S, (const T*)kld, (const T*)add, (const T*)mi, (T*)nll,
                                         (T*)loss, (T*)elbo, B, (int)D, kl_beta, g.G, g.tiles));
  }
  if (loss && means) launch_rows_means<T>(loss, elbo, nll, kld, add, mi, ws, inv_count, means, B, st);
  return hipGetLastError() == hipSuccess ? VAE_OK : VAE_ERR_LAUNCH;
}

template <typename T, int NLL>
static int nll_multi_bwd_dispatch(const void* x, const PtrPack& rp, const PtrPack& gp, bool al, int S,
                                  const void* gnll, const void* gloss, const void* gelbo, const float* gmeans,
                                  void* gkld, void* gadd, void* gmi, int64_t B, int64_t D, float kl_beta,
                                  float inv_count, hipStream_t st) {
  constexpr int VN = VecOf<T>::n;
  const bool vec = (D % VN == 0) && aligned16(x) && al;
  const RowGeom g = nll_geom(B, vec ? D / VN : D, kGridStream, true);
  if (vec) {
    VAE_SWITCH_U(g.U, hipLaunchKernelGGL((nll_multi_bwd_kernel<T, VN, NLL, U>), dim3(g.grid), dim3(kBlock), 0, st,
                                         (const T*)x, rp, gp, S, (const T*)gnll, (const T*)gloss, (const T*)gelbo,
                                         gmeans, (T*)gkld, (T*)gadd, (T*)gmi, B, (int)D, kl_beta, inv_count, g.G,
                                         g.tiles));
  } else {
    VAE_SWITCH_U(g.U, hipLaunchKernelGGL((nll_multi_bwd_kernel<T, 1, NLL, U>), dim3(g.grid), dim3(kBlock), 0, st,
                                         (const T*)x, rp, gp, S, (const T*)gnll, (const T*)gloss, (const T*)gelbo,
                                         gmeans, (T*)gkld, (T*)gadd, (T*)gmi, B, (int)D, kl_beta, inv_count, g.G,
                                         g.tiles));
  }
  return hipGetLastError() == hipSuccess ? VAE_OK : VAE_ERR_LAUNCH;
}

#define NLL_TYPE_SWITCH(T_, FN, ...)                                                     \
  switch (nll_type) {                                                                    \
    case VAE_NLL_BERNOULLI: return FN<T_, VAE_NLL_BERNOULLI>(__VA_ARGS__);               \
    case VAE_NLL_GAUSSIAN: return FN<T_, VAE_NLL_GAUSSIAN>(__VA_ARGS__);                 \
    case VAE_NLL_DISC_LOGISTIC: return FN<T_, VAE_NLL_DISC_LOGISTIC>(__VA_ARGS__);       \
    default: return VAE_ERR_ARG;                                                         \
  }

template <typename T>
static int nll_multi_fwd_type(int nll_type, const void* x, const PtrPack& rp, bool al, int S, const void* kld,
                              const void* add, const void* mi, void* nll, void* loss, void* elbo, float* means,
                              float* ws, int64_t B, int64_t D, float kl_beta, float inv_count, hipStream_t st) {
  NLL_TYPE_SWITCH(T, nll_multi_fwd_dispatch, x, rp, al, S, kld, add, mi, nll, loss, elbo, means, ws, B, D, kl_beta,
                  inv_count, st)
}
template <typename T>
static int nll_multi_bwd_type(int nll_type, const void* x, const PtrPack& rp, const PtrPack& gp, bool al, int S,
                              const void* gnll, const void* gloss, const void* gelbo, const float* gmeans, void* gkld,
                              void* gadd, void* gmi, int64_t B, int64_t D, float kl_beta, float inv_count,
                              hipStream_t st) {
  NLL_TYPE_SWITCH(T, nll_multi_bwd_dispatch, x, rp, gp, al, S, gnll, gloss, gelbo, gmeans, gkld, gadd, gmi, B, D,
                  kl_beta, inv_count, st)
}

}  // namespace vae

using namespace vae;

extern "C" {

int vae_gauss_reparam_multi_fwd(const void* logits, int64_t row_stride, const void* noise, void* z, void* s_out,
                                void* kl, float* means, float* workspace, int64_t B, int64_t L, int64_t S,
                                float eps_tol, float mean_scale, uint64_t seed, uint64_t offset, int64_t row0,
                                int dtype, vae_stream_t stream) {
  if (B < 0 || L <= 0 || L > VAE_MAX_ROW || S < 1 || S > (1 << 20)) return VAE_ERR_ARG;
  if (B == 0) return VAE_OK;
  if (!logits || !z || !kl || !workspace || row_stride < 2 * L) return VAE_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == VAE_F32)
    return gauss_multi_fwd_dispatch<float>(logits, row_stride, noise, z, s_out, kl, means, workspace, B, L, (int)S,
                                           eps_tol, mean_scale, seed, offset, row0, st);
  if (dtype == VAE_BF16)
    return gauss_multi_fwd_dispatch<bf16_t>(logits, row_stride, noise, z, s_out, kl, means, workspace, B, L, (int)S,
                                            eps_tol, mean_scale, seed, offset, row0, st);
  if (dtype == VAE_F16)
    return gauss_multi_fwd_dispatch<f16_t>(logits, row_stride, noise, z, s_out, kl, means, workspace, B, L, (int)S,
                                            eps_tol, mean_scale, seed, offset, row0, st);
  return VAE_ERR_ARG;
}

int vae_gauss_reparam_multi_bwd(const void* logits, int64_t row_stride, const void* noise, const void* gz,
                                const void* gs, const void* gkl, const float* gmeans, void* glogits, int64_t B,
                                int64_t L, int64_t S, float eps_tol, float mean_scale, uint64_t seed, uint64_t offset,
                                int64_t row0, int dtype, vae_stream_t stream) {
  if (B < 0 || L <= 0 || L > VAE_MAX_ROW || S < 1 || S > (1 << 20)) return VAE_ERR_ARG;
  if (B == 0) return VAE_OK;
  if (!logits || !glogits || row_stride < 2 * L) return VAE_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == VAE_F32)
    return gauss_multi_bwd_dispatch<float>(logits, row_stride, noise, gz, gs, gkl, gmeans, glogits, B, L, (int)S,
                                           eps_tol, mean_scale, seed, offset, row0, st);
  if (dtype == VAE_BF16)
    return gauss_multi_bwd_dispatch<bf16_t>(logits, row_stride, noise, gz, gs, gkl, gmeans, glogits, B, L, (int)S,
                                            eps_tol, mean_scale, seed, offset, row0, st);
  if (dtype == VAE_F16)
    return gauss_multi_bwd_dispatch<f16_t>(logits, row_stride, noise, gz, gs, gkl, gmeans, glogits, B, L, (int)S,
                                            eps_tol, mean_scale, seed, offset, row0, st);
  return VAE_ERR_ARG;
}

int vae_nll_multi_elbo_fwd(const void* x, const void* const* recons, int64_t S, int nll_type, const void* kld,
                           const void* add, const void* mi, float kl_beta, float inv_count, void* nll, void* loss,
                           void* elbo, float* means, float* workspace, int64_t B, int64_t D, int dtype,
                           vae_stream_t stream) {
  if (B < 0 || D <= 0 || D > VAE_MAX_ROW || S < 1 || S > VAE_MAX_SAMPLES) return VAE_ERR_ARG;
  if (B == 0) return VAE_OK;
  if (!x || !recons || !nll) return VAE_ERR_ARG;
  if ((loss == nullptr) != (elbo == nullptr)) return VAE_ERR_ARG;
  if (loss && !workspace) return VAE_ERR_ARG;
  PtrPack rp;
  bool al = true;
  if (!pack_ptrs(recons, (int)S, &rp, &al)) return VAE_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == VAE_F32)
    return nll_multi_fwd_type<float>(nll_type, x, rp, al, (int)S, kld, add, mi, nll, loss, elbo, means, workspace, B,
                                     D, kl_beta, inv_count, st);
  if (dtype == VAE_BF16)
    return nll_multi_fwd_type<bf16_t>(nll_type, x, rp, al, (int)S, kld, add, mi, nll, loss, elbo, means, workspace, B,
                                      D, kl_beta, inv_count, st);
  if (dtype == VAE_F16)
    return nll_multi_fwd_type<f16_t>(nll_type, x, rp, al, (int)S, kld, add, mi, nll, loss, elbo, means, workspace, B,
                                      D, kl_beta, inv_count, st);
  return VAE_ERR_ARG;
}

int vae_nll_multi_elbo_bwd(const void* x, const void* const* recons, void* const* grecons, int64_t S, int nll_type,
                           const void* gnll, const void* gloss, const void* gelbo, const float* gmeans, float kl_beta,
                           float inv_count, void* gkld, void* gadd, void* gmi, int64_t B, int64_t D, int dtype,
                           vae_stream_t stream) {
  if (B < 0 || D <= 0 || D > VAE_MAX_ROW || S < 1 || S > VAE_MAX_SAMPLES) return VAE_ERR_ARG;
  if (B == 0) return VAE_OK;
  if (!x || !recons || !grecons) return VAE_ERR_ARG;
  PtrPack rp, gp;
  bool al = true;
  if (!pack_ptrs(recons, (int)S, &rp, &al)) return VAE_ERR_ARG;
  if (!pack_ptrs((const void* const*)grecons, (int)S, &gp, &al)) return VAE_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == VAE_F32)
    return nll_multi_bwd_type<float>(nll_type, x, rp, gp, al, (int)S, gnll, gloss, gelbo, gmeans, gkld, gadd, gmi, B,
                                     D, kl_beta, inv_count, st);
  if (dtype == VAE_BF16)
    return nll_multi_bwd_type<bf16_t>(nll_type, x, rp, gp, al, (int)S, gnll, gloss, gelbo, gmeans, gkld, gadd, gmi,
                                      B, D, kl_beta, inv_count, st);
  if (dtype == VAE_F16)
    return nll_multi_bwd_type<f16_t>(nll_type, x, rp, gp, al, (int)S, gnll, gloss, gelbo, gmeans, gkld, gadd, gmi,
                                      B, D, kl_beta, inv_count, st);
  return VAE_ERR_ARG;
}

}  // extern "C"
