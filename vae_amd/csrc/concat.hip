// ConcatReparameterizer(['isotropic_gaussian', 'discrete']) -- BASELINE config 3's reparameterizer -- as ONE row
// kernel per direction: a group of G lanes owns the whole [2L + K] logits row (contiguous: mu | s | categorical
// logits) and writes the whole [L + K] concatenated sample row.
//
// Replaces, from /root/reference:
//   reparameterizers/concat_reparameterizer.py:110-118  per child: narrow -> child.forward -> .clone(), then torch.cat
//   reparameterizers/isotropic_gaussian.py:77-131,158-170,204-215  (the Gaussian child: see gauss.hip)
//   reparameterizers/gumbel.py:110-124,257-304                     (the Gumbel child: see gumbel_row.hpp)
//   reparameterizers/concat_reparameterizer.py:71-88               KL = sum of the children's (Gumbel's is 0, Q2)
//
// Why one kernel: run child by child (r02) each launch reads a 1 KiB slice out of every 2 KiB row -- two launches
// at ~60 % of the HBM roofline each, and every launch is short enough (17-40 us of traffic) for its fixed costs to
// show.  Here the row is read once, front to back, and the sample row is written once.
// The two mean diagnostics of the Gaussian child (isotropic_gaussian.py:213-214) are NOT taken here: a grid-wide sum
// forces the persistent-grid form with a closing ticket chain (+8-12 us on a 64 us launch, r03); the module class
// produces them on first read with the reference's own torch.mean.
//   training: z_out = [e*(0.5*s) + mu | y]          bytes per sample: 4(2L+K) in, 4(L+K) + 4 out (+ 4L + 4K of
//   eval:     z_out = [mu | z_hard], y and log_q_z  injected noise)
#include "common.hpp"
#include "gumbel_row.hpp"
#include "../../include/vae_amd.h"

namespace vae {

template <typename T, int V, int UG, int UK, bool INJ, bool EXACT>
__global__ __launch_bounds__(kBlock) void concat_gg_fwd_kernel(
    const T* __restrict__ logits, int64_t row_stride, const T* __restrict__ noise, const T* __restrict__ unif,
    T* __restrict__ zcat, int64_t zc_stride, T* __restrict__ y_out, T* __restrict__ lq_out, T* __restrict__ kl,
    int64_t B, int L, int K, float eps_tol, float tau, int training, uint64_t seed_g, uint64_t offset_g,
    uint64_t seed_k, uint64_t offset_k, int64_t row0, int G, int64_t tiles) {
  if constexpr (!INJ) {
    offset_g = resolve_offset(offset_g);
    offset_k = resolve_offset(offset_k);
  }
  const int rpb = kBlock / G;
  const bool nts = nt_stores((int64_t)B * (L + K) * (int64_t)sizeof(T));
  const int sub = threadIdx.x % G;
  const int rloc = threadIdx.x / G;
  const int lvec = L / V, kvec = K / V;

  for (int64_t t = blockIdx.x; t < tiles; t += gridDim.x) {
    const int64_t row = t * rpb + rloc;
    const bool live = row < B;
    const int64_t r = live ? row : B - 1;  // dead lane groups shadow the last row (no stores)
    const T* lrow = logits + r * row_stride;
    T* zrow = zcat + r * zc_stride;

    // the whole row's loads go out before anything is used
    float mu[UG][V], s[UG][V], e[UG][V], l[UK][V], u[UK][V];
    load_chunk<T, V, UG>(lrow, sub, G, lvec, mu);
    load_chunk<T, V, UG>(lrow + L, sub, G, lvec, s);
    load_chunk<T, V, UK>(lrow + 2 * L, sub, G, kvec, l);
    if constexpr (INJ) {
      if (training) load_chunk<T, V, UG>(noise + r * L, sub, G, lvec, e);
      load_chunk<T, V, UK>(unif + r * K, sub, G, kvec, u);
    }

    // ---- Gaussian child (gauss.hip: the same roundings, z bit for bit); chunks past the row end are computed on
    // their clamped copies and masked out of the sum and the stores
    float klr = 0.f;
#pragma unroll
    for (int k = 0; k < UG; ++k) {
      const bool valid = EXACT || sub + k * G < lvec;   // EXACT: lvec == UG * G and kvec == UK * G, no masks
      const int j = valid ? sub + k * G : lvec - 1;
      if constexpr (!INJ) {
        if (training) philox_normal<V>(seed_g, offset_g, (uint64_t)(row0 + r) * (uint64_t)L + (uint64_t)j * V, e[k]);
      }
      float zz[V], klc = 0.f;
#pragma unroll
      for (int i = 0; i < V; ++i) {
        const float si = to_f32(from_f32<T>(add_rn(s[k][i], eps_tol)));
        const float m = mu[k][i];
        zz[i] = training ? add_rn(to_f32(from_f32<T>(mul_rn(e[k][i], to_f32(from_f32<T>(mul_rn(0.5f, si)))))), m) : m;
        klc += 0.5f * (si * si + m * m - 1.0f) - kLn2 * log2_raw(fabsf(si));
      }
      klr += valid ? klc : 0.f;
      if (live && valid) store_vec<T, V>(nts, zrow + (int64_t)j * V, zz);
    }
    klr = group_sum(klr, G);
    if (live && sub == 0) kl[row] = from_f32<T>(klr);

    // ---- Gumbel child: y into the sample row in training, z_hard in eval (gumbel.py:336-341)
    T* yk = training ? zrow + L : (y_out ? y_out + r * K : nullptr);
    T* zh = training ? nullptr : zrow + L;
    gumbel_row_fwd<T, V, UK, INJ, EXACT>(l, u, yk, zh, lq_out ? lq_out + r * K : nullptr, live, nts, sub, G, kvec, tau, seed_k,
                                  offset_k, (uint64_t)(row0 + r) * (uint64_t)K);
  }
}

// Backward of the training-mode forward: d logits [B, 2L + K] (contiguous rows) from the gradient of the sample row
// gz [B, L + K] and of kl [B].  The Gumbel part needs y, which is the sample row's second part.
//   d mu = gz + gkl*mu;  d s = 0.5*e*gz + gkl*(s - 1/s)                           (gauss.hip)
//   d l  = (1/tau) * y * (gy - sum_k gy*y)                                        (gumbel.hip)
template <typename T, int V, int UG, int UK>
__global__ __launch_bounds__(kBlock) void concat_gg_bwd_kernel(
    const T* __restrict__ logits, int64_t row_stride, const T* __restrict__ noise, const T* __restrict__ zcat,
    int64_t zc_stride, const T* __restrict__ gz, int64_t gz_stride, const T* __restrict__ gkl,
    T* __restrict__ glogits, int64_t gl_stride, int64_t B, int L, int K, float eps_tol, float tau, uint64_t seed_g,
    uint64_t offset_g, int64_t row0, int G, int64_t tiles) {
  offset_g = resolve_offset(offset_g);
  const int rpb = kBlock / G;
  const bool nts = nt_stores((int64_t)B * (2 * L + K) * (int64_t)sizeof(T));
  const int sub = threadIdx.x % G;
  const int rloc = threadIdx.x / G;
  const int lvec = L / V, kvec = K / V;
  const float itau = 1.0f / tau;

  for (int64_t t = blockIdx.x; t < tiles; t += gridDim.x) {
    const int64_t row = t * rpb + rloc;
    const bool live = row < B;
    const int64_t r = live ? row : B - 1;
    const T* lrow = logits + r * row_stride;
    T* grow = glogits + r * gl_stride;
    const float gk = gkl ? to_f32(gkl[r]) : 0.f;

    float mu[UG][V], s[UG][V], e[UG][V], g[UG][V], yv[UK][V], gy[UK][V];
    load_chunk<T, V, UG>(lrow, sub, G, lvec, mu);
    load_chunk<T, V, UG>(lrow + L, sub, G, lvec, s);
    if (gz) {
      load_chunk<T, V, UG>(gz + r * gz_stride, sub, G, lvec, g);
      load_chunk<T, V, UK>(gz + r * gz_stride + L, sub, G, kvec, gy);
      load_chunk<T, V, UK>(zcat + r * zc_stride + L, sub, G, kvec, yv);
      if (noise) load_chunk<T, V, UG>(noise + r * L, sub, G, lvec, e);
    }

#pragma unroll
    for (int k = 0; k < UG; ++k) {
      const int j = sub + k * G;
      if (j < lvec) {
        if (gz && !noise)
          philox_normal<V>(seed_g, offset_g, (uint64_t)(row0 + r) * (uint64_t)L + (uint64_t)j * V, e[k]);
        float dmu[V], ds[V];
#pragma unroll
        for (int i = 0; i < V; ++i) {
          const float si = to_f32(from_f32<T>(add_rn(s[k][i], eps_tol)));
          const float gzi = gz ? g[k][i] : 0.f;
          dmu[i] = gzi + gk * mu[k][i];
          ds[i] = (gz ? 0.5f * e[k][i] * gzi : 0.f) + gk * (si - fast_rcp(si));
        }
        if (live) {
          store_vec<T, V>(nts, grow + (int64_t)j * V, dmu);
          store_vec<T, V>(nts, grow + L + (int64_t)j * V, ds);
        }
      }
    }

    float dot = 0.f;
    if (gz) {
#pragma unroll
      for (int k = 0; k < UK; ++k)
        if (sub + k * G < kvec)
#pragma unroll
          for (int i = 0; i < V; ++i) dot += gy[k][i] * yv[k][i];
      dot = row_sum(dot, G);
    }
    if (live) {
#pragma unroll
      for (int k = 0; k < UK; ++k) {
        const int j = sub + k * G;
        if (j < kvec) {
          float o[V];
#pragma unroll
          for (int i = 0; i < V; ++i) o[i] = gz ? itau * yv[k][i] * (gy[k][i] - dot) : 0.f;
          store_vec<T, V>(nts, grow + 2 * L + (int64_t)j * V, o);
        }
      }
    }
  }
}

// Lanes per row: the categorical part stays register-resident (<= 4 chunks per lane, two preferred: r03 sweep of
// the stand-alone kernel, K = 256: G = 32 beats G = 16 by 2-4 points), the Gaussian part then takes ceil(lvec / G)
// chunks, also <= 4.  False when the row does not fit this form (the caller runs the children one by one).
static bool concat_geom(int64_t B, int64_t lvec, int64_t kvec, int64_t max_grid, RowGeom* g, int* UG, int* UK) {
  int G = 1;
  while (G < 64 && ((int64_t)G * 2 < kvec || (int64_t)G * 4 < lvec)) G <<= 1;
  if ((int64_t)G * 4 < kvec || (int64_t)G * 4 < lvec) return false;
  auto chunks = [&](int64_t v) { const int64_t it = (v + G - 1) / G; return it >= 3 ? 4 : (it == 2 ? 2 : 1); };
  *UG = chunks(lvec);
  *UK = chunks(kvec);
  g->G = G;
  g->U = *UK;
  g->rows_per_block = kBlock / G;
  g->tiles = (B + g->rows_per_block - 1) / g->rows_per_block;
  g->grid = (int)(g->tiles < max_grid ? g->tiles : max_grid);
  if (g->grid < 1) g->grid = 1;
  return true;
}

#define VAE_SWITCH_UG_UK(ug_, uk_, ...)                                        \
  switch ((ug_) * 8 + (uk_)) {                                                 \
    case 1 * 8 + 1: { constexpr int UG = 1, UK = 1; __VA_ARGS__; } break;      \
    case 1 * 8 + 2: { constexpr int UG = 1, UK = 2; __VA_ARGS__; } break;      \
    case 1 * 8 + 4: { constexpr int UG = 1, UK = 4; __VA_ARGS__; } break;      \
    case 2 * 8 + 1: { constexpr int UG = 2, UK = 1; __VA_ARGS__; } break;      \
    case 2 * 8 + 2: { constexpr int UG = 2, UK = 2; __VA_ARGS__; } break;      \
    case 2 * 8 + 4: { constexpr int UG = 2, UK = 4; __VA_ARGS__; } break;      \
    case 4 * 8 + 1: { constexpr int UG = 4, UK = 1; __VA_ARGS__; } break;      \
    case 4 * 8 + 2: { constexpr int UG = 4, UK = 2; __VA_ARGS__; } break;      \
    default: { constexpr int UG = 4, UK = 4; __VA_ARGS__; } break;             \
  }

template <typename T>
static int concat_fwd_dispatch(const void* logits, int64_t row_stride, const void* noise, const void* unif, void* zcat,
                               int64_t zc_stride, void* y, void* lq, void* kl, int64_t B, int64_t L, int64_t K,
                               float eps_tol, float tau, int training, uint64_t seed_g, uint64_t offset_g,
                               uint64_t seed_k, uint64_t offset_k, int64_t row0, hipStream_t stream) {
  constexpr int VN = VecOf<T>::n;
  const bool vec = (L % VN == 0) && (K % VN == 0) && (row_stride % VN == 0) && (zc_stride % VN == 0) &&
                   aligned16(logits) && aligned16(zcat) && aligned16(noise) && aligned16(unif) && aligned16(y) &&
                   aligned16(lq);
  RowGeom g;
  int UG, UK;
  if (!vec || !concat_geom(B, L / VN, K / VN, kGridStream, &g, &UG, &UK)) return VAE_ERR_ARG;
  const bool exact = L / VN == (int64_t)UG * g.G && K / VN == (int64_t)UK * g.G;
#define VAE_CONCAT_LAUNCH(INJ_, EX_)                                                                                   \
  VAE_SWITCH_UG_UK(UG, UK, hipLaunchKernelGGL((concat_gg_fwd_kernel<T, VN, UG, UK, INJ_, EX_>), dim3(g.grid),          \
                                              dim3(kBlock), 0, stream, (const T*)logits, row_stride, (const T*)noise,  \
                                              (const T*)unif, (T*)zcat, zc_stride, (T*)y, (T*)lq, (T*)kl, B, (int)L,   \
                                              (int)K, eps_tol, tau, training, seed_g, offset_g, seed_k, offset_k, row0, \
                                              g.G, g.tiles))
  if (unif) {
    if (exact) { VAE_CONCAT_LAUNCH(true, true); } else { VAE_CONCAT_LAUNCH(true, false); }
  } else {
    if (exact) { VAE_CONCAT_LAUNCH(false, true); } else { VAE_CONCAT_LAUNCH(false, false); }
  }
#undef VAE_CONCAT_LAUNCH
  return hipGetLastError() == hipSuccess ? VAE_OK : VAE_ERR_LAUNCH;
}

template <typename T>
static int concat_bwd_dispatch(const void* logits, int64_t row_stride, const void* noise, const void* zcat,
                               int64_t zc_stride, const void* gz, int64_t gz_stride, const void* gkl, void* glogits,
                               int64_t gl_stride, int64_t B, int64_t L, int64_t K, float eps_tol, float tau,
                               uint64_t seed_g, uint64_t offset_g, int64_t row0, hipStream_t stream) {
  constexpr int VN = VecOf<T>::n;
  const bool vec = (L % VN == 0) && (K % VN == 0) && (row_stride % VN == 0) && (zc_stride % VN == 0) &&
                   (gz_stride % VN == 0) && (gl_stride % VN == 0) && aligned16(logits) && aligned16(zcat) &&
                   aligned16(noise) && aligned16(gz) && aligned16(glogits);
  RowGeom g;
  int UG, UK;
  if (!vec || !concat_geom(B, L / VN, K / VN, kGridStream, &g, &UG, &UK)) return VAE_ERR_ARG;
  VAE_SWITCH_UG_UK(UG, UK, hipLaunchKernelGGL((concat_gg_bwd_kernel<T, VN, UG, UK>), dim3(g.grid), dim3(kBlock), 0,
                                              stream, (const T*)logits, row_stride, (const T*)noise, (const T*)zcat,
                                              zc_stride, (const T*)gz, gz_stride, (const T*)gkl, (T*)glogits, gl_stride,
                                              B, (int)L, (int)K, eps_tol, tau, seed_g, offset_g, row0, g.G, g.tiles));
  return hipGetLastError() == hipSuccess ? VAE_OK : VAE_ERR_LAUNCH;
}

}  // namespace vae

using namespace vae;

extern "C" {

int vae_concat_gauss_gumbel_supported(int64_t L, int64_t K, int dtype) {
  const int VN = dtype == VAE_F32 ? 4 : 8;
  if (dtype != VAE_F32 && dtype != VAE_BF16 && dtype != VAE_F16) return 0;
  if (L <= 0 || K <= 0 || L % VN || K % VN) return 0;
  RowGeom g;
  int UG, UK;
  return concat_geom(1, L / VN, K / VN, 1, &g, &UG, &UK) ? 1 : 0;
}

int vae_concat_gauss_gumbel_fwd(const void* logits, int64_t row_stride, const void* noise, const void* unif, void* z_out,
                                int64_t z_stride, void* y, void* log_q_z, void* kl, int64_t B, int64_t L, int64_t K,
                                float eps_tol, float tau, int training, uint64_t seed_gauss, uint64_t offset_gauss,
                                uint64_t seed_gumbel, uint64_t offset_gumbel, int64_t row0, int dtype,
                                vae_stream_t stream) {
  if (B < 0 || L <= 0 || K <= 0 || !(tau > 0.f)) return VAE_ERR_ARG;
  if (B == 0) return VAE_OK;
  if (z_stride == 0) z_stride = L + K;
  if (!logits || !z_out || !kl || row_stride < 2 * L + K || z_stride < L + K) return VAE_ERR_ARG;
  // the noise of the two children comes from one source: both injected (training needs `noise` then), or both drawn
  if (unif ? (training && !noise) : (noise != nullptr)) return VAE_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
#define VAE_CONCAT_FWD(TT)                                                                                            \
  return concat_fwd_dispatch<TT>(logits, row_stride, noise, unif, z_out, z_stride, y, log_q_z, kl, B, L, K, eps_tol, \
                                 tau, training, seed_gauss, offset_gauss, seed_gumbel, offset_gumbel, row0, st)
  if (dtype == VAE_F32) VAE_CONCAT_FWD(float);
  if (dtype == VAE_BF16) VAE_CONCAT_FWD(bf16_t);
  if (dtype == VAE_F16) VAE_CONCAT_FWD(f16_t);
#undef VAE_CONCAT_FWD
  return VAE_ERR_ARG;
}

int vae_concat_gauss_gumbel_bwd(const void* logits, int64_t row_stride, const void* noise, const void* z_out,
                                int64_t z_stride, const void* gz, int64_t gz_stride, const void* gkl, void* glogits,
                                int64_t glogits_stride, int64_t B, int64_t L, int64_t K, float eps_tol, float tau,
                                uint64_t seed_gauss, uint64_t offset_gauss, int64_t row0, int dtype,
                                vae_stream_t stream) {
  if (B < 0 || L <= 0 || K <= 0 || !(tau > 0.f)) return VAE_ERR_ARG;
  if (B == 0) return VAE_OK;
  if (z_stride == 0) z_stride = L + K;
  if (gz_stride == 0) gz_stride = L + K;
  if (glogits_stride == 0) glogits_stride = 2 * L + K;
  if (!logits || !glogits || (gz && !z_out) || row_stride < 2 * L + K || z_stride < L + K || gz_stride < L + K ||
      glogits_stride < 2 * L + K)
    return VAE_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
#define VAE_CONCAT_BWD(TT)                                                                                        \
  return concat_bwd_dispatch<TT>(logits, row_stride, noise, z_out, z_stride, gz, gz_stride, gkl, glogits,   \
                                 glogits_stride, B, L, K, eps_tol, tau, seed_gauss, offset_gauss, row0, st)
  if (dtype == VAE_F32) VAE_CONCAT_BWD(float);
  if (dtype == VAE_BF16) VAE_CONCAT_BWD(bf16_t);
  if (dtype == VAE_F16) VAE_CONCAT_BWD(f16_t);
#undef VAE_CONCAT_BWD
  return VAE_ERR_ARG;
}

}  // extern "C"
