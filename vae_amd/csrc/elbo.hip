// Per-sample NLL + ELBO, fused (forward and backward).
//
// Replaces, from /root/reference:
//   abstract_vae.py:461-471   nll(x, recon_x, nll_type) -> [B]   (delegates to the
//                             un-vendored helpers.distributions.nll; the three
//                             definitions below are this build's stated contract,
//                             "parity unpinned" -- see oracle/cpu_ref.py)
//   abstract_vae.py:356-388   elbo = nll + kld; loss = (nll + add + beta*kld) - mi;
//                             the six torch.mean() calls of the returned dict
//   abstract_vae.py:15-37     VarianceProjector layout: for NLLs with a variance the
//                             decoder row is [mean(D) | variance-param(D)]
//
// HBM-bound: the forward reads 4D (x) + 4D or 8D (recon) bytes per sample and
// writes 12 bytes; the backward reads the same and writes 4D or 8D.  A row is
// owned by G lanes (G <= 64: shuffles only) or by the whole block (G = 256: wave
// shuffles, then LDS across the four waves) when rows are long and few.
#include <mutex>

#include "common.hpp"
#include "nll_math.hpp"
#include "../../include/vae_amd.h"

namespace vae {

// Load policy per kernel (tuning knobs).  Measured in the bench.py sequence (r01): temporal loads in any one of
// the four hot kernels slow that kernel by 25-30 % and only shift ~10 % of time out of its successor; walking the
// backward's rows in reverse (to hit what the forward left in the Infinity Cache) changed nothing -- nt loads do
// not seem to allocate there.  All four stay non-temporal.
#ifndef VAE_NT_NLL_FWD
#define VAE_NT_NLL_FWD 1
#endif
#ifndef VAE_NT_NLL_BWD
#define VAE_NT_NLL_BWD 1
#endif
constexpr bool kNtNllFwd = VAE_NT_NLL_FWD != 0, kNtNllBwd = VAE_NT_NLL_BWD != 0;

// ---------------------------------------------------------------------------
// forward: nll[B] (+ loss[B], elbo[B], block partials of the six sums)
// ---------------------------------------------------------------------------
template <typename T, int V, int NLL, int U, typename Src>
__global__ __launch_bounds__(kBlock) void nll_elbo_fwd_kernel(
    Src src, const T* __restrict__ kld, const T* __restrict__ add, const T* __restrict__ mi, T* __restrict__ nll_out,
    T* __restrict__ loss, T* __restrict__ elbo, int D, float kl_beta, int G, int64_t tiles) {
  const int rpb = kBlock / G;
  const int sub = threadIdx.x % G;
  const int rloc = threadIdx.x / G;
  const int nvec = D / V;
  const int64_t rstride = (NLL == VAE_NLL_BERNOULLI) ? (int64_t)D : 2 * (int64_t)D;

  for (int64_t t = blockIdx.x; t < tiles; t += gridDim.x) {
    int64_t row;
    const T *xr, *rr;
    T* gr_unused;
    const bool live = src.locate(t, rpb, rloc, D, rstride, row, xr, rr, gr_unused);
    const float kb = src.beta(t, kl_beta);
    float s = 0.f;
    float k = 0.f, ad = 0.f, m = 0.f;   // requested with the row (see nll_elbo_fused_kernel)
    if (live) {
      if (loss) {
        if (kld) k = to_f32(kld[row]);
        if (add) ad = to_f32(add[row]);
        if (mi) m = to_f32(mi[row]);
      }
      for (int j0 = sub; j0 < nvec; j0 += G * U) {
        float xv[U][V], a[U][V], b[U][V];
        load_chunk<T, V, U, kNtNllFwd>(xr, j0, G, nvec, xv);
        load_chunk<T, V, U, kNtNllFwd>(rr, j0, G, nvec, a);
        if constexpr (NLL != VAE_NLL_BERNOULLI) load_chunk<T, V, U, kNtNllFwd>(rr + D, j0, G, nvec, b);
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
    s = row_sum(s, G);
    if (live && sub == 0) {
      // round through the tensor dtype where the reference materialises a tensor
      const float n = to_f32(from_f32<T>(s));
      nll_out[row] = from_f32<T>(n);
      if (loss) {
        const float e = to_f32(from_f32<T>(n + k));
        const float l = to_f32(from_f32<T>((n + ad + kb * k) - m));
        loss[row] = from_f32<T>(l);
        elbo[row] = from_f32<T>(e);
      }
    }
  }
}

// ---------------------------------------------------------------------------
// backward: d recon (and the [B]-sized d kld / d add / d mi)
//   c_nll[b] = gnll[b] + gloss[b] + gelbo[b] + (gm_loss + gm_elbo + gm_nll) * inv_count
// ---------------------------------------------------------------------------
template <typename T, int V, int NLL, int U, typename Src>
__global__ __launch_bounds__(kBlock) void nll_elbo_bwd_kernel(
    Src src, const T* __restrict__ gnll, const T* __restrict__ gloss, const T* __restrict__ gelbo,
    const float* __restrict__ gmeans, T* __restrict__ gkld, T* __restrict__ gadd, T* __restrict__ gmi,
    int64_t rows_total, int D, float kl_beta, float inv_count, int G, int64_t tiles, int cpr) {
  const int rpb = kBlock / G;
  const bool nts = nt_stores(rows_total * D * (int64_t)sizeof(T));
  const int sub = threadIdx.x % G;
  const int rloc = threadIdx.x / G;
  const int nvec = D / V;
  const int64_t rstride = (NLL == VAE_NLL_BERNOULLI) ? (int64_t)D : 2 * (int64_t)D;
  float gm[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) gm[k] = gmeans ? gmeans[k] * inv_count : 0.f;

  // cpr > 1 (block-per-row geometry, long rows): a row is cut into cpr chunks of G * U vectors and every block takes
  // ONE chunk -- the backward needs no row sum, so nothing ties the chunks of a row together.  One pass of loads,
  // arithmetic and stores per block instead of a loop whose every iteration first waits for the acknowledgement of
  // the previous iteration's stores (vmcnt retires in order): 3x64x64 rows 5.33 -> TB/s of the flat 3R2W stream
  // (scripts/membench 402 10 w: 6.06 TB/s).
  for (int64_t tc = blockIdx.x; tc < tiles * cpr; tc += gridDim.x) {
    const int64_t t = cpr == 1 ? tc : tc / cpr;
    const int chunk = cpr == 1 ? 0 : (int)(tc - t * cpr);
    int64_t row;
    const T *xr, *rr;
    T* gr;
    if (!src.locate(t, rpb, rloc, D, rstride, row, xr, rr, gr)) continue;
    const float gl = (gloss ? to_f32(gloss[row]) : 0.f) + gm[0];
    const float ge = (gelbo ? to_f32(gelbo[row]) : 0.f) + gm[1];
    const float c = (gnll ? to_f32(gnll[row]) : 0.f) + gl + ge + gm[2];
    if (sub == 0 && chunk == 0) {
      if (gkld) gkld[row] = from_f32<T>(src.beta(t, kl_beta) * gl + ge + gm[3]);
      if (gadd) gadd[row] = from_f32<T>(gl + gm[4]);
      if (gmi) gmi[row] = from_f32<T>(gm[5] - gl);
    }
    const int jend = cpr == 1 ? nvec : min(nvec, (chunk + 1) * G * U);
    for (int j0 = sub + chunk * G * U; j0 < jend; j0 += G * U) {
      float xv[U][V], a[U][V], b[U][V];
      load_chunk<T, V, U, kNtNllBwd>(xr, j0, G, nvec, xv);
      load_chunk<T, V, U, kNtNllBwd>(rr, j0, G, nvec, a);
      if constexpr (NLL != VAE_NLL_BERNOULLI) load_chunk<T, V, U, kNtNllBwd>(rr + D, j0, G, nvec, b);
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

// ---------------------------------------------------------------------------
// ELBO combine from a given nll[B] (K>1 Monte-Carlo average, VRNN per-step dicts):
// one thread per sample.
// ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(kBlock) void elbo_combine_fwd_kernel(
    const T* __restrict__ nll, const T* __restrict__ kld, const T* __restrict__ add, const T* __restrict__ mi,
    T* __restrict__ loss, T* __restrict__ elbo, float* __restrict__ ws, float inv_count, float* __restrict__ means,
    int64_t B, float kl_beta) {
  float acc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int64_t row = (int64_t)blockIdx.x * kBlock + threadIdx.x; row < B; row += (int64_t)gridDim.x * kBlock) {
    const float n = to_f32(nll[row]);
    const float k = kld ? to_f32(kld[row]) : 0.f;
    const float ad = add ? to_f32(add[row]) : 0.f;
    const float m = mi ? to_f32(mi[row]) : 0.f;
    const float e = to_f32(from_f32<T>(n + k));
    const float l = to_f32(from_f32<T>((n + ad + kl_beta * k) - m));
    loss[row] = from_f32<T>(l);
    elbo[row] = from_f32<T>(e);
    acc[0] += l; acc[1] += e; acc[2] += n; acc[3] += k; acc[4] += ad; acc[5] += m;
  }
  if (ws) grid_reduce<6>(acc, ws, inv_count, means, kNanElbo6);
}

template <typename T>
__global__ __launch_bounds__(kBlock) void elbo_combine_bwd_kernel(
    const T* __restrict__ gloss, const T* __restrict__ gelbo, const float* __restrict__ gmeans, T* __restrict__ gnll,
    T* __restrict__ gkld, T* __restrict__ gadd, T* __restrict__ gmi, int64_t B, float kl_beta, float inv_count) {
  float gm[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) gm[k] = gmeans ? gmeans[k] * inv_count : 0.f;
  for (int64_t row = (int64_t)blockIdx.x * kBlock + threadIdx.x; row < B; row += (int64_t)gridDim.x * kBlock) {
    const float gl = (gloss ? to_f32(gloss[row]) : 0.f) + gm[0];
    const float ge = (gelbo ? to_f32(gelbo[row]) : 0.f) + gm[1];
    if (gnll) gnll[row] = from_f32<T>(gl + ge + gm[2]);
    if (gkld) gkld[row] = from_f32<T>(kl_beta * gl + ge + gm[3]);
    if (gadd) gadd[row] = from_f32<T>(gl + gm[4]);
    if (gmi) gmi[row] = from_f32<T>(gm[5] - gl);
  }
}


// ---------------------------------------------------------------------------
// forward + gradient in ONE pass over x and recon (no reference counterpart: the reference's
// autograd re-reads both tensors in its backward).  Besides everything the forward kernel
// writes, grecon = cscale * d nll / d recon is stored, cscale = inv_count * grad_scale: exactly
// d(loss_mean)/d(recon) when the step backpropagates loss_mean alone (loss['loss_mean'].backward(),
// the reference's training loop).  The step then moves 12D (20D for the variance NLLs) bytes per
// sample through HBM instead of 8D + 12D (12D + 20D).  vae_nll_fused_grad_finish rescales the
// stored gradient on the device when the upstream gradient turns out not to be 1.
// ---------------------------------------------------------------------------
template <typename T, int V, int NLL, int U>
__global__ __launch_bounds__(kBlock) void nll_elbo_fused_kernel(
    const T* __restrict__ x, const T* __restrict__ recon, const T* __restrict__ kld, const T* __restrict__ add,
    const T* __restrict__ mi, T* __restrict__ nll_out, T* __restrict__ loss, T* __restrict__ elbo,
    T* __restrict__ grecon, float cscale, int64_t B, int D, float kl_beta, int G, int64_t tiles) {
  const int rpb = kBlock / G;
  const bool nts = nt_stores((int64_t)B * D * (int64_t)sizeof(T));
  const int sub = threadIdx.x % G;
  const int rloc = threadIdx.x / G;
  const int nvec = D / V;
  const int64_t rstride = (NLL == VAE_NLL_BERNOULLI) ? (int64_t)D : 2 * (int64_t)D;

  for (int64_t t = blockIdx.x; t < tiles; t += gridDim.x) {
    const int64_t row = t * rpb + rloc;
    float s = 0.f;
    f32x2 s2 = {0.f, 0.f};   // Bernoulli: the packed pair of partial sums (nll_bern_value_grad)
    // the [B] inputs of the row epilogue, requested with the row itself (behind the row sum they were a dependent
    // round trip with nothing else of the wave in flight)
    float k = 0.f, ad = 0.f, m = 0.f;
    if (row < B) {
      if (kld) k = to_f32(kld[row]);
      if (add) ad = to_f32(add[row]);
      if (mi) m = to_f32(mi[row]);
      const T* xr = x + row * D;
      const T* rr = recon + row * rstride;
      T* gr = grecon + row * rstride;
      __builtin_assume(sub < nvec);   // every geometry gives a lane at least one vector: no zero-trip path
      for (int j0 = sub; j0 < nvec; j0 += G * U) {
        float xv[U][V], a[U][V], b[U][V];
        load_chunk<T, V, U, kNtNllFwd>(xr, j0, G, nvec, xv);
        load_chunk<T, V, U, kNtNllFwd>(rr, j0, G, nvec, a);
        if constexpr (NLL != VAE_NLL_BERNOULLI) load_chunk<T, V, U, kNtNllFwd>(rr + D, j0, G, nvec, b);
        // take delivery of the epilogue's inputs HERE, where only loads are younger: vmcnt retires in order, so a
        // wait for them behind the gradient stores is a wait for the acknowledgement of every one of those stores
        asm volatile("" ::"v"(k), "v"(ad), "v"(m));
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int j = j0 + u * G;
          if (j < nvec) {
            float ga[V], gb[V];
            if constexpr (NLL == VAE_NLL_BERNOULLI) nll_bern_value_grad<V>(xv[u], a[u], cscale, s2, ga);
#pragma unroll
            for (int i = 0; i < V; ++i) {
              if constexpr (NLL == VAE_NLL_GAUSSIAN) {
                s += nll_gauss(xv[u][i], a[u][i], b[u][i]);
                nll_gauss_grad(xv[u][i], a[u][i], b[u][i], ga[i], gb[i]);
                ga[i] *= cscale; gb[i] *= cscale;
              }
              if constexpr (NLL == VAE_NLL_DISC_LOGISTIC) {
                s += nll_dlog(xv[u][i], a[u][i], b[u][i]);
                nll_dlog_grad(xv[u][i], a[u][i], b[u][i], ga[i], gb[i]);
                ga[i] *= cscale; gb[i] *= cscale;
              }
            }
            store_vec<T, V>(nts, gr + (int64_t)j * V, ga);
            if constexpr (NLL != VAE_NLL_BERNOULLI) store_vec<T, V>(nts, gr + D + (int64_t)j * V, gb);
          }
        }
      }
    }
    if constexpr (NLL == VAE_NLL_BERNOULLI) s = s2[0] + s2[1];
    s = row_sum(s, G);
    if (row < B && sub == 0) {
      const float n = to_f32(from_f32<T>(s));
      const float e = to_f32(from_f32<T>(n + k));
      const float l = to_f32(from_f32<T>((n + ad + kl_beta * k) - m));
      nll_out[row] = from_f32<T>(n);
      loss[row] = from_f32<T>(l);
      elbo[row] = from_f32<T>(e);
    }
  }
}

// The same for long rows (block-per-row geometry), one CHUNK of kBlock * U vectors per block: a block-per-row loop
// waits, at every iteration, for the acknowledgement of the previous iteration's gradient stores before its next
// loads return (loads and stores retire through one in-order counter), and measured 5.2-5.3 TB/s on 3x64x64 rows
// where the flat 3R2W stream runs 6.06 (scripts/membench 402 10 w).  Here every block makes one pass -- loads,
// arithmetic, stores -- and leaves the sum of its chunk in parts[row * cpr + chunk]; nll_rows_from_parts_kernel adds
// a row's chunks in chunk order (deterministic) and finishes the row.
template <typename T, int V, int NLL, int U>
__global__ __launch_bounds__(kBlock) void nll_elbo_fused_chunk_kernel(
    const T* __restrict__ x, const T* __restrict__ recon, T* __restrict__ grecon, float* __restrict__ parts,
    float cscale, int64_t B, int D, int cpr) {
  const bool nts = nt_stores((int64_t)B * D * (int64_t)sizeof(T));
  const int nvec = D / V;
  const int64_t rstride = (NLL == VAE_NLL_BERNOULLI) ? (int64_t)D : 2 * (int64_t)D;
  for (int64_t tc = blockIdx.x; tc < B * cpr; tc += gridDim.x) {
    const int64_t row = tc / cpr;
    const int chunk = (int)(tc - row * cpr);
    const T* xr = x + row * D;
    const T* rr = recon + row * rstride;
    T* gr = grecon + row * rstride;
    const int j0 = threadIdx.x + chunk * kBlock * U;
    float s = 0.f;
    f32x2 s2 = {0.f, 0.f};
    float xv[U][V], a[U][V], b[U][V];
    load_chunk<T, V, U, kNtNllFwd>(xr, j0, kBlock, nvec, xv);
    load_chunk<T, V, U, kNtNllFwd>(rr, j0, kBlock, nvec, a);
    if constexpr (NLL != VAE_NLL_BERNOULLI) load_chunk<T, V, U, kNtNllFwd>(rr + D, j0, kBlock, nvec, b);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int j = j0 + u * kBlock;
      if (j < nvec) {
        float ga[V], gb[V];
        if constexpr (NLL == VAE_NLL_BERNOULLI) nll_bern_value_grad<V>(xv[u], a[u], cscale, s2, ga);
#pragma unroll
        for (int i = 0; i < V; ++i) {
          if constexpr (NLL == VAE_NLL_GAUSSIAN) {
            s += nll_gauss(xv[u][i], a[u][i], b[u][i]);
            nll_gauss_grad(xv[u][i], a[u][i], b[u][i], ga[i], gb[i]);
            ga[i] *= cscale; gb[i] *= cscale;
          }
          if constexpr (NLL == VAE_NLL_DISC_LOGISTIC) {
            s += nll_dlog(xv[u][i], a[u][i], b[u][i]);
            nll_dlog_grad(xv[u][i], a[u][i], b[u][i], ga[i], gb[i]);
            ga[i] *= cscale; gb[i] *= cscale;
          }
        }
        store_vec<T, V>(nts, gr + (int64_t)j * V, ga);
        if constexpr (NLL != VAE_NLL_BERNOULLI) store_vec<T, V>(nts, gr + D + (int64_t)j * V, gb);
      }
    }
    if constexpr (NLL == VAE_NLL_BERNOULLI) s = s2[0] + s2[1];
    s = row_sum(s, kBlock);
    if (threadIdx.x == 0) parts[tc] = s;
  }
}

// nll / loss / elbo [B] from the per-chunk sums: one thread per row, chunks added in chunk order.
template <typename T>
__global__ __launch_bounds__(kBlock) void nll_rows_from_parts_kernel(
    const float* __restrict__ parts, int cpr, const T* __restrict__ kld, const T* __restrict__ add,
    const T* __restrict__ mi, T* __restrict__ nll_out, T* __restrict__ loss, T* __restrict__ elbo, int64_t B,
    float kl_beta) {
  for (int64_t row = (int64_t)blockIdx.x * kBlock + threadIdx.x; row < B; row += (int64_t)gridDim.x * kBlock) {
    float s = parts[row * cpr];
    for (int c = 1; c < cpr; ++c) s += parts[row * cpr + c];
    const float k = kld ? to_f32(kld[row]) : 0.f;
    const float ad = add ? to_f32(add[row]) : 0.f;
    const float m = mi ? to_f32(mi[row]) : 0.f;
    const float n = to_f32(from_f32<T>(s));
    nll_out[row] = from_f32<T>(n);
    loss[row] = from_f32<T>((n + ad + kl_beta * k) - m);
    elbo[row] = from_f32<T>(n + k);
  }
}

// buf *= (gmeans[0] + gmeans[1] + gmeans[2]) unless that sum is exactly 1 (then no memory is touched):
// the stored d(loss_mean)/d(recon) becomes the gradient for any upstream weights on loss_mean, elbo_mean, nll_mean.
template <typename T, int V>
__global__ __launch_bounds__(kBlock) void scale_by_mean_grads_kernel(T* __restrict__ buf, const float* __restrict__ gmeans,
                                                                     int64_t nvec_total) {
  const float c = gmeans[0] + gmeans[1] + gmeans[2];
  if (c == 1.0f) return;
  for (int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x; j < nvec_total; j += (int64_t)gridDim.x * kBlock) {
    float v[V];
    load_vec<T, V, false>(buf + j * V, v);
#pragma unroll
    for (int i = 0; i < V; ++i) v[i] *= c;
    store_vec<T, V>(buf + j * V, v);
  }
}

// The whole backward of a one-pass step whose upstream gradients are the six mean gradients only, in ONE launch:
// the [B]-sized d kld / d add / d mi (constants per row: elbo_combine_bwd with no per-sample upstream) and the
// conditional rescale of the stored d recon.
template <typename T, int V>
__global__ __launch_bounds__(kBlock) void nll_fused_backward_kernel(T* __restrict__ buf, const float* __restrict__ gmeans,
                                                                    int64_t nvec_total, T* __restrict__ gkld,
                                                                    T* __restrict__ gadd, T* __restrict__ gmi, int64_t B,
                                                                    float kl_beta, float inv_count) {
  float gm[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) gm[k] = gmeans[k] * inv_count;
  const T vk = from_f32<T>(kl_beta * gm[0] + gm[1] + gm[3]);
  const T va = from_f32<T>(gm[0] + gm[4]);
  const T vm = from_f32<T>(gm[5] - gm[0]);
  for (int64_t row = (int64_t)blockIdx.x * kBlock + threadIdx.x; row < B; row += (int64_t)gridDim.x * kBlock) {
    if (gkld) gkld[row] = vk;
    if (gadd) gadd[row] = va;
    if (gmi) gmi[row] = vm;
  }
  const float c = gmeans[0] + gmeans[1] + gmeans[2];
  if (c == 1.0f) return;
  for (int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x; j < nvec_total; j += (int64_t)gridDim.x * kBlock) {
    float v[V];
    load_vec<T, V, false>(buf + j * V, v);
#pragma unroll
    for (int i = 0; i < V; ++i) v[i] *= c;
    store_vec<T, V>(buf + j * V, v);
  }
}

template <typename T, int NLL>
static int nll_fused_dispatch(const void* x, const void* recon, const void* kld, const void* add, const void* mi,
                              void* nll, void* loss, void* elbo, float* means, float* workspace, void* grecon,
                              float cscale, int64_t B, int64_t D, float kl_beta, float inv_count, hipStream_t stream,
                              bool has_row_parts) {
  constexpr int VN = VecOf<T>::n;
  const bool vec = (D % VN == 0) && aligned16(x) && aligned16(recon) && aligned16(grecon);
  const RowGeom g = nll_geom(B, vec ? D / VN : D, kGridStream, true);
  const int64_t nvec = vec ? D / VN : D;
  const int64_t cpr = (nvec + (int64_t)kBlock * 4 - 1) / ((int64_t)kBlock * 4);
  if (vec && has_row_parts && g.G == kBlock && cpr > 1 && B * cpr <= (int64_t)kRowPartFloats && !knobs().maxgrid) {
    // long rows: one chunk per block, per-chunk sums behind the workspace's counters and block partials
    float* parts = workspace + kWorkspaceFloats;
    const int64_t work = B * cpr;
    const int grid = (int)(work < kGridStream ? work : kGridStream);
    hipLaunchKernelGGL((nll_elbo_fused_chunk_kernel<T, VN, NLL, 4>), dim3(grid), dim3(kBlock), 0, stream, (const T*)x,
                       (const T*)recon, (T*)grecon, parts, cscale, B, (int)D, (int)cpr);
    const int64_t rb = (B + kBlock - 1) / kBlock;
    hipLaunchKernelGGL((nll_rows_from_parts_kernel<T>), dim3((int)(rb < 1024 ? rb : 1024)), dim3(kBlock), 0, stream,
                       parts, (int)cpr, (const T*)kld, (const T*)add, (const T*)mi, (T*)nll, (T*)loss, (T*)elbo, B,
                       kl_beta);
    launch_rows_means<T>(loss, elbo, nll, kld, add, mi, workspace, inv_count, means, B, stream);
    return hipGetLastError() == hipSuccess ? VAE_OK : VAE_ERR_LAUNCH;
  }
  if (vec) {
    VAE_SWITCH_U(g.U, hipLaunchKernelGGL((nll_elbo_fused_kernel<T, VN, NLL, U>), dim3(g.grid), dim3(kBlock), 0, stream,
                                         (const T*)x, (const T*)recon, (const T*)kld, (const T*)add, (const T*)mi,
                                         (T*)nll, (T*)loss, (T*)elbo, (T*)grecon, cscale, B, (int)D, kl_beta, g.G,
                                         g.tiles));
  } else {
    VAE_SWITCH_U(g.U, hipLaunchKernelGGL((nll_elbo_fused_kernel<T, 1, NLL, U>), dim3(g.grid), dim3(kBlock), 0, stream,
                                         (const T*)x, (const T*)recon, (const T*)kld, (const T*)add, (const T*)mi,
                                         (T*)nll, (T*)loss, (T*)elbo, (T*)grecon, cscale, B, (int)D, kl_beta, g.G,
                                         g.tiles));
  }
  launch_rows_means<T>(loss, elbo, nll, kld, add, mi, workspace, inv_count, means, B, stream);
  return hipGetLastError() == hipSuccess ? VAE_OK : VAE_ERR_LAUNCH;
}

template <typename T>
static int nll_fused_type(int nll_type, const void* x, const void* recon, const void* kld, const void* add,
                          const void* mi, void* nll, void* loss, void* elbo, float* means, float* workspace,
                          void* grecon, float cscale, int64_t B, int64_t D, float kl_beta, float inv_count,
                          hipStream_t st, bool has_row_parts) {
  switch (nll_type) {
    case VAE_NLL_BERNOULLI:
      return nll_fused_dispatch<T, VAE_NLL_BERNOULLI>(x, recon, kld, add, mi, nll, loss, elbo, means, workspace, grecon,
                                                      cscale, B, D, kl_beta, inv_count, st, has_row_parts);
    case VAE_NLL_GAUSSIAN:
      return nll_fused_dispatch<T, VAE_NLL_GAUSSIAN>(x, recon, kld, add, mi, nll, loss, elbo, means, workspace, grecon,
                                                     cscale, B, D, kl_beta, inv_count, st, has_row_parts);
    case VAE_NLL_DISC_LOGISTIC:
      return nll_fused_dispatch<T, VAE_NLL_DISC_LOGISTIC>(x, recon, kld, add, mi, nll, loss, elbo, means, workspace,
                                                          grecon, cscale, B, D, kl_beta, inv_count, st, has_row_parts);
  }
  return VAE_ERR_ARG;
}

// launch of the two general kernels over any row source (FlatRows / StepRows)
template <typename T, int NLL, typename Src>
static void nll_fwd_launch(const Src& src, bool vec, const RowGeom& g, int64_t tiles, const void* kld, const void* add,
                           const void* mi, void* nll, void* loss, void* elbo, int64_t D, float kl_beta,
                           hipStream_t stream) {
  constexpr int VN = VecOf<T>::n;
  const int grid = (int)(tiles < kGridStream ? tiles : kGridStream);
  if (vec) {
    VAE_SWITCH_U(g.U, hipLaunchKernelGGL((nll_elbo_fwd_kernel<T, VN, NLL, U, Src>), dim3(grid), dim3(kBlock), 0, stream,
                                         src, (const T*)kld, (const T*)add, (const T*)mi, (T*)nll, (T*)loss, (T*)elbo,
                                         (int)D, kl_beta, g.G, tiles));
  } else {
    VAE_SWITCH_U(g.U, hipLaunchKernelGGL((nll_elbo_fwd_kernel<T, 1, NLL, U, Src>), dim3(grid), dim3(kBlock), 0, stream,
                                         src, (const T*)kld, (const T*)add, (const T*)mi, (T*)nll, (T*)loss, (T*)elbo,
                                         (int)D, kl_beta, g.G, tiles));
  }
}

template <typename T, int NLL, typename Src>
static void nll_bwd_launch(const Src& src, bool vec, const RowGeom& g, int64_t tiles, int64_t rows_total,
                           const void* gnll, const void* gloss, const void* gelbo, const float* gmeans, void* gkld,
                           void* gadd, void* gmi, int64_t D, float kl_beta, float inv_count, hipStream_t stream) {
  constexpr int VN = VecOf<T>::n;
  // block-per-row geometry: one chunk of G * U vectors per block (see the kernel)
  const int64_t nvec = vec ? D / VN : D;
  const int cpr = (g.G == kBlock && !knobs().maxgrid) ? (int)((nvec + (int64_t)kBlock * g.U - 1) / ((int64_t)kBlock * g.U)) : 1;
  const int64_t work = tiles * cpr;
  const int grid = (int)(work < kGridStream ? work : kGridStream);
  if (vec) {
    VAE_SWITCH_U(g.U, hipLaunchKernelGGL((nll_elbo_bwd_kernel<T, VN, NLL, U, Src>), dim3(grid), dim3(kBlock), 0, stream,
                                         src, (const T*)gnll, (const T*)gloss, (const T*)gelbo, gmeans, (T*)gkld,
                                         (T*)gadd, (T*)gmi, rows_total, (int)D, kl_beta, inv_count, g.G, tiles, cpr));
  } else {
    VAE_SWITCH_U(g.U, hipLaunchKernelGGL((nll_elbo_bwd_kernel<T, 1, NLL, U, Src>), dim3(grid), dim3(kBlock), 0, stream,
                                         src, (const T*)gnll, (const T*)gloss, (const T*)gelbo, gmeans, (T*)gkld,
                                         (T*)gadd, (T*)gmi, rows_total, (int)D, kl_beta, inv_count, g.G, tiles, cpr));
  }
}

template <typename T, int NLL>
static int nll_fwd_dispatch(const void* x, const void* recon, const void* kld, const void* add, const void* mi,
                            void* nll, void* loss, void* elbo, float* means, float* workspace, int64_t B, int64_t D,
                            float kl_beta, float inv_count, hipStream_t stream) {
  constexpr int VN = VecOf<T>::n;
  const bool vec = (D % VN == 0) && aligned16(x) && aligned16(recon);
  const RowGeom g = nll_geom(B, vec ? D / VN : D, kGridStream, false, true);
  const FlatRows<T> src{(const T*)x, (const T*)recon, nullptr, B};
  nll_fwd_launch<T, NLL>(src, vec, g, g.tiles, kld, add, mi, nll, loss, elbo, D, kl_beta, stream);
  if (loss && means) launch_rows_means<T>(loss, elbo, nll, kld, add, mi, workspace, inv_count, means, B, stream);
  return hipGetLastError() == hipSuccess ? VAE_OK : VAE_ERR_LAUNCH;
}

template <typename T, int NLL>
static int nll_bwd_dispatch(const void* x, const void* recon, const void* gnll, const void* gloss, const void* gelbo,
                            const float* gmeans, void* grecon, void* gkld, void* gadd, void* gmi, int64_t B,
                            int64_t D, float kl_beta, float inv_count, hipStream_t stream) {
  constexpr int VN = VecOf<T>::n;
  const bool vec = (D % VN == 0) && aligned16(x) && aligned16(recon) && aligned16(grecon);
  const RowGeom g = nll_geom(B, vec ? D / VN : D, kGridStream, true);
  const FlatRows<T> src{(const T*)x, (const T*)recon, (T*)grecon, B};
  nll_bwd_launch<T, NLL>(src, vec, g, g.tiles, B, gnll, gloss, gelbo, gmeans, gkld, gadd, gmi, D, kl_beta, inv_count,
                         stream);
  return hipGetLastError() == hipSuccess ? VAE_OK : VAE_ERR_LAUNCH;
}

// ---- the same over T per-step tensors given by pointer (VRNN / AdditiveVRNN loss_function) ---------------------
template <typename Tab>
static bool fill_steps(Tab* tab, const void* const* xs, const void* const* recons, void* const* grecons, int64_t steps,
                       const float* kl_betas, float kl_beta, bool* aligned) {
  for (int64_t i = 0; i < steps; ++i) {
    if (!xs[i] || !recons[i] || (grecons && !grecons[i])) return false;
    tab->x[i] = xs[i];
    tab->recon[i] = recons[i];
    tab->beta[i] = kl_betas ? kl_betas[i] : kl_beta;
    if (!aligned16(xs[i]) || !aligned16(recons[i]) || (grecons && !aligned16(grecons[i]))) *aligned = false;
    if constexpr (sizeof(Tab) == sizeof(StepTableG)) tab->grecon[i] = grecons[i];
  }
  return true;
}

template <typename T, int NLL>
static int nll_steps_fwd_dispatch(const void* const* xs, const void* const* recons, int64_t steps, const void* kld,
                                  const void* add, const void* mi, const float* kl_betas, void* nll, void* loss,
                                  void* elbo, void* loss_avg, void* elbo_avg, void* elbo_lse, float* means,
                                  float* workspace, int64_t B, int64_t D, float kl_beta, float inv_count,
                                  hipStream_t stream) {
  constexpr int VN = VecOf<T>::n;
  StepRows<T, StepTable> src;
  bool al = true;
  if (!fill_steps(&src.tab, xs, recons, nullptr, steps, kl_betas, kl_beta, &al)) return VAE_ERR_ARG;
  const bool vec = (D % VN == 0) && al;
  const RowGeom g = nll_geom(B * steps, vec ? D / VN : D, kGridStream, false, true);   // geometry by the whole launch
  src.B = B;
  src.tps = (B + g.rows_per_block - 1) / g.rows_per_block;                // tiles never straddle steps
  nll_fwd_launch<T, NLL>(src, vec, g, src.tps * steps, kld, add, mi, nll, loss, elbo, D, kl_beta, stream);
  launch_rows_means<T>(loss, elbo, nll, kld, add, mi, workspace, inv_count, means, B * steps, stream, steps, loss_avg,
                       elbo_avg, elbo_lse);
  return hipGetLastError() == hipSuccess ? VAE_OK : VAE_ERR_LAUNCH;
}

template <typename T, int NLL>
static int nll_steps_bwd_dispatch(const void* const* xs, const void* const* recons, void* const* grecons, int64_t steps,
                                  const void* gnll, const void* gloss, const void* gelbo, const float* gmeans,
                                  const float* kl_betas, void* gkld, void* gadd, void* gmi, int64_t B, int64_t D,
                                  float kl_beta, float inv_count, hipStream_t stream) {
  constexpr int VN = VecOf<T>::n;
  StepRows<T, StepTableG> src;
  bool al = true;
  if (!fill_steps(&src.tab, xs, recons, grecons, steps, kl_betas, kl_beta, &al)) return VAE_ERR_ARG;
  const bool vec = (D % VN == 0) && al;
  const RowGeom g = nll_geom(B * steps, vec ? D / VN : D, kGridStream, true);
  src.B = B;
  src.tps = (B + g.rows_per_block - 1) / g.rows_per_block;
  nll_bwd_launch<T, NLL>(src, vec, g, src.tps * steps, B * steps, gnll, gloss, gelbo, gmeans, gkld, gadd, gmi, D,
                         kl_beta, inv_count, stream);
  return hipGetLastError() == hipSuccess ? VAE_OK : VAE_ERR_LAUNCH;
}

template <typename T>
static int nll_fwd_type(int nll_type, const void* x, const void* recon, const void* kld, const void* add,
                        const void* mi, void* nll, void* loss, void* elbo, float* means, float* workspace, int64_t B,
                        int64_t D, float kl_beta, float inv_count, hipStream_t st) {
  switch (nll_type) {
    case VAE_NLL_BERNOULLI:
      return nll_fwd_dispatch<T, VAE_NLL_BERNOULLI>(x, recon, kld, add, mi, nll, loss, elbo, means, workspace, B, D,
                                                    kl_beta, inv_count, st);
    case VAE_NLL_GAUSSIAN:
      return nll_fwd_dispatch<T, VAE_NLL_GAUSSIAN>(x, recon, kld, add, mi, nll, loss, elbo, means, workspace, B, D,
                                                   kl_beta, inv_count, st);
    case VAE_NLL_DISC_LOGISTIC:
      return nll_fwd_dispatch<T, VAE_NLL_DISC_LOGISTIC>(x, recon, kld, add, mi, nll, loss, elbo, means, workspace, B,
                                                        D, kl_beta, inv_count, st);
  }
  return VAE_ERR_ARG;
}

template <typename T>
static int nll_bwd_type(int nll_type, const void* x, const void* recon, const void* gnll, const void* gloss,
                        const void* gelbo, const float* gmeans, void* grecon, void* gkld, void* gadd, void* gmi,
                        int64_t B, int64_t D, float kl_beta, float inv_count, hipStream_t st) {
  switch (nll_type) {
    case VAE_NLL_BERNOULLI:
      return nll_bwd_dispatch<T, VAE_NLL_BERNOULLI>(x, recon, gnll, gloss, gelbo, gmeans, grecon, gkld, gadd, gmi, B,
                                                    D, kl_beta, inv_count, st);
    case VAE_NLL_GAUSSIAN:
      return nll_bwd_dispatch<T, VAE_NLL_GAUSSIAN>(x, recon, gnll, gloss, gelbo, gmeans, grecon, gkld, gadd, gmi, B,
                                                   D, kl_beta, inv_count, st);
    case VAE_NLL_DISC_LOGISTIC:
      return nll_bwd_dispatch<T, VAE_NLL_DISC_LOGISTIC>(x, recon, gnll, gloss, gelbo, gmeans, grecon, gkld, gadd,
                                                        gmi, B, D, kl_beta, inv_count, st);
  }
  return VAE_ERR_ARG;
}

}  // namespace vae

// ---------------------------------------------------------------------------
// Scratch for vae_nll_elbo_fused_static: what `workspace` is to the other entry points (the closing launch's
// arrival counter and block partials), held by the code object instead of the caller.  Each GPU the library runs
// on has its own copy at its own address; `scratch` itself is a host-side stand-in for the symbol, never an address
// to hand to a kernel -- host code uses the pointer hipGetSymbolAddress returned for the current device.
// ---------------------------------------------------------------------------
static __device__ __align__(256) float scratch[vae::kWorkspaceFloats];

namespace {
constexpr int kMaxDevices = 64;
std::mutex g_scratch_mutex;
float* g_scratch_ptr[kMaxDevices];   // per device, filled on that device's first call

// The current device's copy of `scratch` (looked up once per device, with that device current).
int scratch_for_current_device(float** out) {
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return VAE_ERR_LAUNCH;
  std::lock_guard<std::mutex> lock(g_scratch_mutex);
  if (!g_scratch_ptr[dev]) {
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)) != hipSuccess || !p) return VAE_ERR_LAUNCH;
    g_scratch_ptr[dev] = static_cast<float*>(p);
  }
  *out = g_scratch_ptr[dev];
  return VAE_OK;
}
}  // namespace

using namespace vae;

extern "C" {

int vae_nll_elbo_fwd(const void* x, const void* recon, int nll_type, const void* kld, const void* add, const void* mi,
                     float kl_beta, float inv_count, void* nll, void* loss, void* elbo, float* means,
                     float* workspace, int64_t B, int64_t D, int dtype, vae_stream_t stream) {
  if (B < 0 || D <= 0 || D > VAE_MAX_ROW) return VAE_ERR_ARG;
  if (B == 0) return VAE_OK;
  if (!x || !recon || !nll) return VAE_ERR_ARG;
  if ((loss == nullptr) != (elbo == nullptr)) return VAE_ERR_ARG;
  if (loss && !workspace) return VAE_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == VAE_F32)
    return nll_fwd_type<float>(nll_type, x, recon, kld, add, mi, nll, loss, elbo, means, workspace, B, D, kl_beta,
                               inv_count, st);
  if (dtype == VAE_BF16)
    return nll_fwd_type<bf16_t>(nll_type, x, recon, kld, add, mi, nll, loss, elbo, means, workspace, B, D, kl_beta,
                                inv_count, st);
  if (dtype == VAE_F16)
    return nll_fwd_type<f16_t>(nll_type, x, recon, kld, add, mi, nll, loss, elbo, means, workspace, B, D, kl_beta,
                                inv_count, st);
  return VAE_ERR_ARG;
}

int vae_nll_elbo_bwd(const void* x, const void* recon, int nll_type, const void* gnll, const void* gloss,
                     const void* gelbo, const float* gmeans, float kl_beta, float inv_count, void* grecon, void* gkld,
                     void* gadd, void* gmi, int64_t B, int64_t D, int dtype, vae_stream_t stream) {
  if (B < 0 || D <= 0 || D > VAE_MAX_ROW) return VAE_ERR_ARG;
  if (B == 0) return VAE_OK;
  if (!x || !recon || !grecon) return VAE_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == VAE_F32)
    return nll_bwd_type<float>(nll_type, x, recon, gnll, gloss, gelbo, gmeans, grecon, gkld, gadd, gmi, B, D,
                               kl_beta, inv_count, st);
  if (dtype == VAE_BF16)
    return nll_bwd_type<bf16_t>(nll_type, x, recon, gnll, gloss, gelbo, gmeans, grecon, gkld, gadd, gmi, B, D,
                                kl_beta, inv_count, st);
  if (dtype == VAE_F16)
    return nll_bwd_type<f16_t>(nll_type, x, recon, gnll, gloss, gelbo, gmeans, grecon, gkld, gadd, gmi, B, D,
                                kl_beta, inv_count, st);
  return VAE_ERR_ARG;
}

int vae_nll_steps_elbo_fwd(const void* const* xs, const void* const* recons, int64_t steps, int nll_type,
                           const void* kld, const void* add, const void* mi, const float* kl_betas, float kl_beta,
                           float inv_count, void* nll, void* loss, void* elbo, void* loss_avg, void* elbo_avg,
                           void* elbo_lse, float* means, float* workspace, int64_t B, int64_t D, int dtype,
                           vae_stream_t stream) {
  if (B < 0 || D <= 0 || D > VAE_MAX_ROW || steps < 1 || steps > VAE_MAX_STEPS) return VAE_ERR_ARG;
  if (B == 0) return VAE_OK;
  if (!xs || !recons || !nll || !loss || !elbo || !means || !workspace) return VAE_ERR_ARG;
  if ((loss_avg == nullptr) != (elbo_avg == nullptr)) return VAE_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
#define VAE_STEPS_FWD(TT, NN)                                                                                        \
  return nll_steps_fwd_dispatch<TT, NN>(xs, recons, steps, kld, add, mi, kl_betas, nll, loss, elbo, loss_avg,        \
                                        elbo_avg, elbo_lse, means, workspace, B, D, kl_beta, inv_count, st)
#define VAE_STEPS_FWD_T(TT)                                       \
  switch (nll_type) {                                             \
    case VAE_NLL_BERNOULLI: VAE_STEPS_FWD(TT, VAE_NLL_BERNOULLI); \
    case VAE_NLL_GAUSSIAN: VAE_STEPS_FWD(TT, VAE_NLL_GAUSSIAN);   \
    case VAE_NLL_DISC_LOGISTIC: VAE_STEPS_FWD(TT, VAE_NLL_DISC_LOGISTIC); \
  }                                                               \
  return VAE_ERR_ARG
  if (dtype == VAE_F32) { VAE_STEPS_FWD_T(float); }
  if (dtype == VAE_BF16) { VAE_STEPS_FWD_T(bf16_t); }
  if (dtype == VAE_F16) { VAE_STEPS_FWD_T(f16_t); }
#undef VAE_STEPS_FWD_T
#undef VAE_STEPS_FWD
  return VAE_ERR_ARG;
}

int vae_nll_steps_elbo_bwd(const void* const* xs, const void* const* recons, void* const* grecons, int64_t steps,
                           int nll_type, const void* gnll, const void* gloss, const void* gelbo, const float* gmeans,
                           const float* kl_betas, float kl_beta, float inv_count, void* gkld, void* gadd, void* gmi,
                           int64_t B, int64_t D, int dtype, vae_stream_t stream) {
  if (B < 0 || D <= 0 || D > VAE_MAX_ROW || steps < 1 || steps > VAE_MAX_STEPS) return VAE_ERR_ARG;
  if (B == 0) return VAE_OK;
  if (!xs || !recons || !grecons) return VAE_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
#define VAE_STEPS_BWD(TT, NN)                                                                                       \
  return nll_steps_bwd_dispatch<TT, NN>(xs, recons, grecons, steps, gnll, gloss, gelbo, gmeans, kl_betas, gkld,     \
                                        gadd, gmi, B, D, kl_beta, inv_count, st)
#define VAE_STEPS_BWD_T(TT)                                       \
  switch (nll_type) {                                             \
    case VAE_NLL_BERNOULLI: VAE_STEPS_BWD(TT, VAE_NLL_BERNOULLI); \
    case VAE_NLL_GAUSSIAN: VAE_STEPS_BWD(TT, VAE_NLL_GAUSSIAN);   \
    case VAE_NLL_DISC_LOGISTIC: VAE_STEPS_BWD(TT, VAE_NLL_DISC_LOGISTIC); \
  }                                                               \
  return VAE_ERR_ARG
  if (dtype == VAE_F32) { VAE_STEPS_BWD_T(float); }
  if (dtype == VAE_BF16) { VAE_STEPS_BWD_T(bf16_t); }
  if (dtype == VAE_F16) { VAE_STEPS_BWD_T(f16_t); }
#undef VAE_STEPS_BWD_T
#undef VAE_STEPS_BWD
  return VAE_ERR_ARG;
}

static int nll_elbo_fused_impl(const void* x, const void* recon, int nll_type, const void* kld, const void* add,
                               const void* mi, float kl_beta, float inv_count, float grad_scale, void* nll, void* loss,
                               void* elbo, float* means, float* workspace, void* grecon, int64_t B, int64_t D, int dtype,
                               vae_stream_t stream, bool has_row_parts) {
  if (B < 0 || D <= 0 || D > VAE_MAX_ROW) return VAE_ERR_ARG;
  if (B == 0) return VAE_OK;
  if (!x || !recon || !nll || !loss || !elbo || !means || !workspace || !grecon) return VAE_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  const float cscale = inv_count * grad_scale;
  if (dtype == VAE_F32)
    return nll_fused_type<float>(nll_type, x, recon, kld, add, mi, nll, loss, elbo, means, workspace, grecon, cscale,
                                 B, D, kl_beta, inv_count, st, has_row_parts);
  if (dtype == VAE_BF16)
    return nll_fused_type<bf16_t>(nll_type, x, recon, kld, add, mi, nll, loss, elbo, means, workspace, grecon, cscale,
                                  B, D, kl_beta, inv_count, st, has_row_parts);
  if (dtype == VAE_F16)
    return nll_fused_type<f16_t>(nll_type, x, recon, kld, add, mi, nll, loss, elbo, means, workspace, grecon, cscale,
                                  B, D, kl_beta, inv_count, st, has_row_parts);
  return VAE_ERR_ARG;
}

int vae_nll_elbo_fused(const void* x, const void* recon, int nll_type, const void* kld, const void* add, const void* mi,
                       float kl_beta, float inv_count, float grad_scale, void* nll, void* loss, void* elbo, float* means,
                       float* workspace, void* grecon, int64_t B, int64_t D, int dtype, vae_stream_t stream) {
  return nll_elbo_fused_impl(x, recon, nll_type, kld, add, mi, kl_beta, inv_count, grad_scale, nll, loss, elbo, means,
                             workspace, grecon, B, D, dtype, stream, true);
}

int vae_nll_elbo_fused_static(const void* x, const void* recon, int nll_type, const void* kld, const void* add,
                              const void* mi, float kl_beta, float inv_count, float grad_scale, void* nll, void* loss,
                              void* elbo, float* means, void* grecon, int64_t B, int64_t D, int dtype,
                              vae_stream_t stream) {
  if (B < 0 || B > VAE_STATIC_MAX_ROWS || D <= 0 || D > VAE_MAX_ROW) return VAE_ERR_ARG;
  if (B == 0) return VAE_OK;
  float* ws = nullptr;
  const int rc = scratch_for_current_device(&ws);
  if (rc != VAE_OK) return rc;
  // The array keeps the last call's contents: the arrival counters must read zero when the closing launch starts
  // (a completed call leaves them at zero; one that was cut short need not).  The saw-a-NaN word behind them stays.
  if (hipMemsetAsync(ws, 0, vae::kWsNanFlag * sizeof(float), (hipStream_t)stream) != hipSuccess) return VAE_ERR_LAUNCH;
  // the library's own array has no per-chunk area: long rows keep the block-per-row loop here
  return nll_elbo_fused_impl(x, recon, nll_type, kld, add, mi, kl_beta, inv_count, grad_scale, nll, loss, elbo, means,
                             ws, grecon, B, D, dtype, stream, false);
}

int vae_nan_flag(float* workspace, int clear, uint32_t* flags_out, vae_stream_t stream) {
  if (!flags_out) return VAE_ERR_ARG;
  float* ws = workspace;
  if (!ws) {
    const int rc = scratch_for_current_device(&ws);
    if (rc != VAE_OK) return rc;
  }
  hipStream_t st = (hipStream_t)stream;
  if (hipMemcpyAsync(flags_out, ws + vae::kWsNanFlag, sizeof(uint32_t), hipMemcpyDeviceToHost, st) != hipSuccess)
    return VAE_ERR_LAUNCH;
  if (clear && hipMemsetAsync(ws + vae::kWsNanFlag, 0, sizeof(uint32_t), st) != hipSuccess) return VAE_ERR_LAUNCH;
  return hipStreamSynchronize(st) == hipSuccess ? VAE_OK : VAE_ERR_LAUNCH;
}

int vae_nll_fused_grad_finish(void* grecon, const float* gmeans, int64_t n, int dtype, vae_stream_t stream) {
  if (n < 0 || (dtype != VAE_F32 && dtype != VAE_BF16 && dtype != VAE_F16)) return VAE_ERR_ARG;
  if (n == 0) return VAE_OK;
  if (!grecon || !gmeans) return VAE_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  const int VN = dtype == VAE_F32 ? 4 : 8;
  const bool vec = (n % VN == 0) && aligned16(grecon);
  const int64_t items = vec ? n / VN : n;
  int64_t blocks = (items + kBlock - 1) / kBlock;
  const int grid = (int)(blocks < 4096 ? blocks : 4096);
  if (dtype == VAE_F32) {
    if (vec) hipLaunchKernelGGL((scale_by_mean_grads_kernel<float, 4>), dim3(grid), dim3(kBlock), 0, st, (float*)grecon, gmeans, items);
    else hipLaunchKernelGGL((scale_by_mean_grads_kernel<float, 1>), dim3(grid), dim3(kBlock), 0, st, (float*)grecon, gmeans, items);
  } else if (dtype == VAE_BF16) {
    if (vec) hipLaunchKernelGGL((scale_by_mean_grads_kernel<bf16_t, 8>), dim3(grid), dim3(kBlock), 0, st, (bf16_t*)grecon, gmeans, items);
    else hipLaunchKernelGGL((scale_by_mean_grads_kernel<bf16_t, 1>), dim3(grid), dim3(kBlock), 0, st, (bf16_t*)grecon, gmeans, items);
  } else {
    if (vec) hipLaunchKernelGGL((scale_by_mean_grads_kernel<f16_t, 8>), dim3(grid), dim3(kBlock), 0, st, (f16_t*)grecon, gmeans, items);
    else hipLaunchKernelGGL((scale_by_mean_grads_kernel<f16_t, 1>), dim3(grid), dim3(kBlock), 0, st, (f16_t*)grecon, gmeans, items);
  }
  return hipGetLastError() == hipSuccess ? VAE_OK : VAE_ERR_LAUNCH;
}

int vae_nll_fused_backward(void* grecon, int64_t n, const float* gmeans, float kl_beta, float inv_count, void* gkld,
                           void* gadd, void* gmi, int64_t B, int dtype, vae_stream_t stream) {
  if (n < 0 || B < 0 || (dtype != VAE_F32 && dtype != VAE_BF16 && dtype != VAE_F16)) return VAE_ERR_ARG;
  if (n == 0 && B == 0) return VAE_OK;
  if (!gmeans || (n > 0 && !grecon)) return VAE_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  const int VN = dtype == VAE_F32 ? 4 : 8;
  const bool vec = (n % VN == 0) && aligned16(grecon);
  const int64_t items = vec ? n / VN : n;
  // sized for the [B] part; the (rare) rescale walks the gradient with the same grid
  int64_t blocks = (B + kBlock - 1) / kBlock;
  if (blocks < 1024 && items > B) blocks = (items + kBlock - 1) / kBlock < 1024 ? (items + kBlock - 1) / kBlock : 1024;
  const int grid = (int)(blocks < 4096 ? (blocks < 1 ? 1 : blocks) : 4096);
  if (dtype == VAE_F32) {
    if (vec) hipLaunchKernelGGL((nll_fused_backward_kernel<float, 4>), dim3(grid), dim3(kBlock), 0, st, (float*)grecon, gmeans, items, (float*)gkld, (float*)gadd, (float*)gmi, B, kl_beta, inv_count);
    else hipLaunchKernelGGL((nll_fused_backward_kernel<float, 1>), dim3(grid), dim3(kBlock), 0, st, (float*)grecon, gmeans, items, (float*)gkld, (float*)gadd, (float*)gmi, B, kl_beta, inv_count);
  } else if (dtype == VAE_BF16) {
    if (vec) hipLaunchKernelGGL((nll_fused_backward_kernel<bf16_t, 8>), dim3(grid), dim3(kBlock), 0, st, (bf16_t*)grecon, gmeans, items, (bf16_t*)gkld, (bf16_t*)gadd, (bf16_t*)gmi, B, kl_beta, inv_count);
    else hipLaunchKernelGGL((nll_fused_backward_kernel<bf16_t, 1>), dim3(grid), dim3(kBlock), 0, st, (bf16_t*)grecon, gmeans, items, (bf16_t*)gkld, (bf16_t*)gadd, (bf16_t*)gmi, B, kl_beta, inv_count);
  } else {
    if (vec) hipLaunchKernelGGL((nll_fused_backward_kernel<f16_t, 8>), dim3(grid), dim3(kBlock), 0, st, (f16_t*)grecon, gmeans, items, (f16_t*)gkld, (f16_t*)gadd, (f16_t*)gmi, B, kl_beta, inv_count);
    else hipLaunchKernelGGL((nll_fused_backward_kernel<f16_t, 1>), dim3(grid), dim3(kBlock), 0, st, (f16_t*)grecon, gmeans, items, (f16_t*)gkld, (f16_t*)gadd, (f16_t*)gmi, B, kl_beta, inv_count);
  }
  return hipGetLastError() == hipSuccess ? VAE_OK : VAE_ERR_LAUNCH;
}

int vae_elbo_combine_fwd(const void* nll, const void* kld, const void* add, const void* mi, float kl_beta,
                         float inv_count, void* loss, void* elbo, float* means, float* workspace, int64_t B,
                         int dtype, vae_stream_t stream) {
  if (B < 0) return VAE_ERR_ARG;
  if (B == 0) return VAE_OK;
  if (!nll || !loss || !elbo || !workspace) return VAE_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  int64_t blocks = (B + kBlock - 1) / kBlock;
  const int grid = (int)(blocks < kReduceGridMax ? blocks : kReduceGridMax);
  if (dtype == VAE_F32) {
    hipLaunchKernelGGL((elbo_combine_fwd_kernel<float>), dim3(grid), dim3(kBlock), 0, st, (const float*)nll,
                       (const float*)kld, (const float*)add, (const float*)mi, (float*)loss, (float*)elbo,
                       means ? workspace : nullptr, inv_count, means, B, kl_beta);
  } else if (dtype == VAE_BF16) {
    hipLaunchKernelGGL((elbo_combine_fwd_kernel<bf16_t>), dim3(grid), dim3(kBlock), 0, st, (const bf16_t*)nll,
                       (const bf16_t*)kld, (const bf16_t*)add, (const bf16_t*)mi, (bf16_t*)loss, (bf16_t*)elbo,
                       means ? workspace : nullptr, inv_count, means, B, kl_beta);
  } else if (dtype == VAE_F16) {
    hipLaunchKernelGGL((elbo_combine_fwd_kernel<f16_t>), dim3(grid), dim3(kBlock), 0, st, (const f16_t*)nll,
                       (const f16_t*)kld, (const f16_t*)add, (const f16_t*)mi, (f16_t*)loss, (f16_t*)elbo,
                       means ? workspace : nullptr, inv_count, means, B, kl_beta);
  } else {
    return VAE_ERR_ARG;
  }
  return hipGetLastError() == hipSuccess ? VAE_OK : VAE_ERR_LAUNCH;
}

int vae_elbo_combine_bwd(const void* gloss, const void* gelbo, const float* gmeans, float kl_beta, float inv_count,
                         void* gnll, void* gkld, void* gadd, void* gmi, int64_t B, int dtype, vae_stream_t stream) {
  if (B < 0) return VAE_ERR_ARG;
  if (B == 0) return VAE_OK;
  hipStream_t st = (hipStream_t)stream;
  int64_t blocks = (B + kBlock - 1) / kBlock;
  const int grid = (int)(blocks < kReduceGridMax ? blocks : kReduceGridMax);
  if (dtype == VAE_F32) {
    hipLaunchKernelGGL((elbo_combine_bwd_kernel<float>), dim3(grid), dim3(kBlock), 0, st, (const float*)gloss,
                       (const float*)gelbo, gmeans, (float*)gnll, (float*)gkld, (float*)gadd, (float*)gmi, B, kl_beta,
                       inv_count);
  } else if (dtype == VAE_BF16) {
    hipLaunchKernelGGL((elbo_combine_bwd_kernel<bf16_t>), dim3(grid), dim3(kBlock), 0, st, (const bf16_t*)gloss,
                       (const bf16_t*)gelbo, gmeans, (bf16_t*)gnll, (bf16_t*)gkld, (bf16_t*)gadd, (bf16_t*)gmi, B,
                       kl_beta, inv_count);
  } else if (dtype == VAE_F16) {
    hipLaunchKernelGGL((elbo_combine_bwd_kernel<f16_t>), dim3(grid), dim3(kBlock), 0, st, (const f16_t*)gloss,
                       (const f16_t*)gelbo, gmeans, (f16_t*)gnll, (f16_t*)gkld, (f16_t*)gadd, (f16_t*)gmi, B,
                       kl_beta, inv_count);
  } else {
    return VAE_ERR_ARG;
  }
  return hipGetLastError() == hipSuccess ? VAE_OK : VAE_ERR_LAUNCH;
}

}  // extern "C"
