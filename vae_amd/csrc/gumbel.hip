// Gumbel-softmax sample (soft + straight-through hard) and the categorical KL, fused.
//
// Replaces the op chains of /root/reference reparameterizers/gumbel.py:
//   :119      log_q_z = log_softmax(logits)
//   :269-271  U = torch.rand (on the CPU, then an H2D copy of [B,K] every step);
//             g = -log(-log(U + 1e-9) + 1e-9)
//   :276-277  y = softmax((logits + g)/tau + 1e-9)
//   :296-301  h = (y == rowmax(y))  (multi-hot on ties);  z_hard = (h - y).detach() + y
//   :222-254  KL = sum softmax(a) * (log_softmax(a) - log_softmax(b)), a/b = log_q_z of q / prior
//
// The whole row lives in registers (K <= 4 * 256 vectors): one read of logits and U, one write
// of y, z_hard and log_q_z -- 20K + 4 bytes per sample instead of ~10 launches over [B,K].
// The hard mask is computed from y exactly as the reference does (same op order: x - max, exp,
// times 1/sum as ATen's CPU softmax, == rowmax), so ties give the same multi-hot rows.
#include "common.hpp"
#include "gumbel_row.hpp"
#include "../../include/vae_amd.h"

namespace vae {

// ---------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------
template <typename T, int V, int U, bool INJ>
__global__ __launch_bounds__(kBlock) void gumbel_fwd_kernel(
    const T* __restrict__ logits, int64_t row_stride, const T* __restrict__ unif, T* __restrict__ y_out,
    int64_t y_stride, T* __restrict__ zh_out, int64_t zh_stride, T* __restrict__ lq_out, int64_t B, int K, float tau, uint64_t seed, uint64_t offset,
    int64_t row0, int G, int64_t tiles) {
  offset = resolve_offset(offset);
  const int rpb = kBlock / G;
  const bool nts = nt_stores((int64_t)B * K * (int64_t)sizeof(T));
  const int sub = threadIdx.x % G;
  const int rloc = threadIdx.x / G;
  const int nvec = K / V;

  for (int64_t t = blockIdx.x; t < tiles; t += gridDim.x) {
    const int64_t row = t * rpb + rloc;
    const bool live = row < B;
    const int64_t r = live ? row : B - 1;  // dead lane groups shadow the last row (no stores)
    float l[U][V], u[U][V];
    load_chunk<T, V, U>(logits + r * row_stride, sub, G, nvec, l);
    if constexpr (INJ) {
      if (y_out || zh_out) load_chunk<T, V, U>(unif + r * K, sub, G, nvec, u);
    }
    gumbel_row_fwd<T, V, U, INJ>(l, u, y_out ? y_out + r * y_stride : nullptr,
                                 zh_out ? zh_out + r * zh_stride : nullptr, lq_out ? lq_out + r * K : nullptr, live,
                                 nts, sub, G, nvec, tau, seed, offset, (uint64_t)(row0 + r) * (uint64_t)K);
  }
}

// ---------------------------------------------------------------------------
// backward: g_logits = (1/tau) * y * (gt - sum(gt*y)) + (glq - exp(lq) * sum(glq)),
// gt = gy + gzh (the straight-through hard sample passes its gradient to y unchanged)
// ---------------------------------------------------------------------------
template <typename T, int V, int U>
__global__ __launch_bounds__(kBlock) void gumbel_bwd_kernel(
    const T* __restrict__ y, int64_t y_stride, const T* __restrict__ lq, const T* __restrict__ gy, int64_t gy_stride,
    const T* __restrict__ gzh, int64_t gzh_stride, const T* __restrict__ glq, T* __restrict__ glogits,
    int64_t gl_stride, int64_t B, int K, float tau, int G, int64_t tiles) {
  const int rpb = kBlock / G;
  const bool nts = nt_stores((int64_t)B * K * (int64_t)sizeof(T));
  const int sub = threadIdx.x % G;
  const int rloc = threadIdx.x / G;
  const int nvec = K / V;
  const float itau = 1.0f / tau;

  for (int64_t t = blockIdx.x; t < tiles; t += gridDim.x) {
    const int64_t row = t * rpb + rloc;
    const bool live = row < B;
    const int64_t r = live ? row : B - 1;
    float yv[U][V], gt[U][V], q[U][V], gq[U][V];
    const bool soft = gy || gzh;
    if (soft) load_chunk<T, V, U>(y + r * y_stride, sub, G, nvec, yv);
    if (gy) load_chunk<T, V, U>(gy + r * gy_stride, sub, G, nvec, gt);
    if (gzh) load_chunk<T, V, U>(gzh + r * gzh_stride, sub, G, nvec, q);
    if (gy && gzh) {
#pragma unroll
      for (int k = 0; k < U; ++k)
#pragma unroll
        for (int i = 0; i < V; ++i) gt[k][i] += q[k][i];
    } else if (gzh) {
#pragma unroll
      for (int k = 0; k < U; ++k)
#pragma unroll
        for (int i = 0; i < V; ++i) gt[k][i] = q[k][i];
    }
    if (glq) {
      load_chunk<T, V, U>(lq + r * K, sub, G, nvec, q);
      load_chunk<T, V, U>(glq + r * K, sub, G, nvec, gq);
    }
    float dot = 0.f, sg = 0.f;
#pragma unroll
    for (int k = 0; k < U; ++k)
      if (sub + k * G < nvec)
#pragma unroll
        for (int i = 0; i < V; ++i) {
          if (soft) dot += gt[k][i] * yv[k][i];
          if (glq) sg += gq[k][i];
        }
    if (soft) dot = row_sum(dot, G);
    if (glq) sg = row_sum(sg, G);
    if (live) {
#pragma unroll
      for (int k = 0; k < U; ++k) {
        const int j = sub + k * G;
        if (j < nvec) {
          float o[V];
#pragma unroll
          for (int i = 0; i < V; ++i) {
            float v = 0.f;
            if (soft) v += itau * yv[k][i] * (gt[k][i] - dot);
            if (glq) v += gq[k][i] - exp_acc(q[k][i]) * sg;
            o[i] = v;
          }
          store_vec<T, V>(nts, glogits + row * gl_stride + (int64_t)j * V, o);
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------
// categorical KL: kl = sum_k p_k (la_k - lb_k), la = log_softmax(a), lb = log_softmax(b), p = exp(la).
// OneHotCategorical(logits=.) normalises once more (logits - logsumexp): idempotent up to
// rounding on an already normalised log_q_z, folded into the one log_softmax here.
// ---------------------------------------------------------------------------
// a -> log_softmax(a) in place; p receives softmax(a) from the SAME exponentials (exp(a - max) / sum: one exp per
// element instead of a second exp(log_softmax)).
template <typename T, int V, int U>
__device__ __forceinline__ void row_log_softmax(float (&a)[U][V], float (&p)[U][V], int sub, int G, int nvec) {
  float m = kNegInf;
#pragma unroll
  for (int k = 0; k < U; ++k)
    if (sub + k * G < nvec)
#pragma unroll
      for (int i = 0; i < V; ++i) m = fmaxf(m, a[k][i]);
  m = row_max(m, G);
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < U; ++k)
#pragma unroll
    for (int i = 0; i < V; ++i) {
      p[k][i] = exp_acc(a[k][i] - m);
      if (sub + k * G < nvec) s += p[k][i];
    }
  s = row_sum(s, G);
  const float lse = m + logf(s);
  const float inv = rcp_nr(s);
#pragma unroll
  for (int k = 0; k < U; ++k)
#pragma unroll
    for (int i = 0; i < V; ++i) {
      a[k][i] -= lse;
      p[k][i] *= inv;
    }
}

template <typename T, int V, int U, bool BWD>
__global__ __launch_bounds__(kBlock) void categorical_kl_kernel(
    const T* __restrict__ a, int64_t stride_a, const T* __restrict__ b, int64_t stride_b, const T* __restrict__ gkl,
    T* __restrict__ kl, T* __restrict__ ga, T* __restrict__ gb, int64_t B, int K, int G, int64_t tiles) {
  const int rpb = kBlock / G;
  const bool nts = nt_stores((int64_t)B * K * (int64_t)sizeof(T));
  const int sub = threadIdx.x % G;
  const int rloc = threadIdx.x / G;
  const int nvec = K / V;
  for (int64_t t = blockIdx.x; t < tiles; t += gridDim.x) {
    const int64_t row = t * rpb + rloc;
    const bool live = row < B;
    const int64_t r = live ? row : B - 1;
    float la[U][V], lb[U][V];
    load_chunk<T, V, U>(a + r * stride_a, sub, G, nvec, la);
    load_chunk<T, V, U>(b + r * stride_b, sub, G, nvec, lb);
    float pa[U][V], pb[U][V];
    row_log_softmax<T, V, U>(la, pa, sub, G, nvec);
    row_log_softmax<T, V, U>(lb, pb, sub, G, nvec);
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < U; ++k)
      if (sub + k * G < nvec)
#pragma unroll
        for (int i = 0; i < V; ++i) s += pa[k][i] * (la[k][i] - lb[k][i]);
    s = row_sum(s, G);
    if (!live) continue;
    if constexpr (!BWD) {
      if (sub == 0) kl[row] = from_f32<T>(s);
    } else {
      const float gk = to_f32(gkl[row]);
#pragma unroll
      for (int k = 0; k < U; ++k) {
        const int j = sub + k * G;
        if (j < nvec) {
          float oa[V], ob[V];
#pragma unroll
          for (int i = 0; i < V; ++i) {
            const float p = pa[k][i];
            oa[i] = gk * p * ((la[k][i] - lb[k][i]) - s);
            ob[i] = gk * (pb[k][i] - p);
          }
          store_vec<T, V>(nts, ga + row * K + (int64_t)j * V, oa);
          if (gb) store_vec<T, V>(nts, gb + row * K + (int64_t)j * V, ob);
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------
// Rows too long for the register-resident kernels above (more than 4 * 256 vectors: K > 4096 fp32, 8192 in the
// 16-bit types; the reference has no limit, gumbel.py:110-124): one block per row, the row streamed three times
// (maxima, sums, outputs).  A row is at most 64 MiB and normally a few hundred KiB, so the second and third pass
// come out of L2 / Infinity Cache; the arithmetic is the resident kernels', element for element.
// ---------------------------------------------------------------------------
template <typename T, int V>
__device__ __forceinline__ void gumbel_logit(const float (&l)[V], float (&u)[V], bool injected, float tau,
                                             float (&x)[V]) {
#pragma unroll
  for (int i = 0; i < V; ++i) {
    float g;
    if (injected) {
      g = -logf(add_rn(u[i], kGumbelEps));
      g = -logf(add_rn(g, kGumbelEps));
    } else {
      g = -fast_log(u[i] + kGumbelEps);
      g = -fast_log(g + kGumbelEps);
    }
    g = to_f32(from_f32<T>(g));
    const float xi = to_f32(from_f32<T>(div_rn(to_f32(from_f32<T>(add_rn(l[i], g))), tau)));
    x[i] = to_f32(from_f32<T>(add_rn(xi, kGumbelEps)));
  }
}

template <typename T, int V>
__global__ __launch_bounds__(kBlock) void gumbel_fwd_long_kernel(
    const T* __restrict__ logits, int64_t row_stride, const T* __restrict__ unif, T* __restrict__ y_out,
    int64_t y_stride, T* __restrict__ zh_out, int64_t zh_stride, T* __restrict__ lq_out, int64_t B, int K, float tau,
    uint64_t seed, uint64_t offset, int64_t row0) {
  offset = resolve_offset(offset);
  const int nvec = K / V;
  const bool want_lq = lq_out != nullptr;
  const bool want_soft = y_out != nullptr || zh_out != nullptr;
  for (int64_t row = blockIdx.x; row < B; row += gridDim.x) {
    const T* lr = logits + row * row_stride;
    const T* ur = unif ? unif + row * (int64_t)K : nullptr;
    auto logit_x = [&](int j, const float (&l)[V], float (&x)[V]) {
      float u[V];
      if (ur) load_vec<T, V>(ur + (int64_t)j * V, u);
      else philox_uniform<V>(seed, offset, (uint64_t)(row0 + row) * (uint64_t)K + (uint64_t)j * V, u);
      gumbel_logit<T, V>(l, u, ur != nullptr, tau, x);
    };
    float m1 = kNegInf, m2 = kNegInf;
    for (int j = threadIdx.x; j < nvec; j += kBlock) {
      float l[V], x[V];
      load_vec<T, V>(lr + (int64_t)j * V, l);
#pragma unroll
      for (int i = 0; i < V; ++i) m1 = fmaxf(m1, l[i]);
      if (want_soft) {
        logit_x(j, l, x);
#pragma unroll
        for (int i = 0; i < V; ++i) m2 = fmaxf(m2, x[i]);
      }
    }
    m1 = row_max(m1, kBlock);
    if (want_soft) m2 = row_max(m2, kBlock);
    float s1 = 0.f, s2 = 0.f;
    for (int j = threadIdx.x; j < nvec; j += kBlock) {
      float l[V], x[V];
      load_vec<T, V>(lr + (int64_t)j * V, l);
      if (want_lq)
#pragma unroll
        for (int i = 0; i < V; ++i) s1 += exp_acc(sub_rn(l[i], m1));
      if (want_soft) {
        logit_x(j, l, x);
#pragma unroll
        for (int i = 0; i < V; ++i) s2 += expf(sub_rn(x[i], m2));
      }
    }
    float lse = 0.f, rinv = 0.f, ymax = 0.f;
    if (want_lq) lse = logf(row_sum(s1, kBlock));
    if (want_soft) {
      rinv = div_rn(1.0f, row_sum(s2, kBlock));
      ymax = to_f32(from_f32<T>(rinv));   // the largest y is the one at x = max: exp(0) * rinv, rounded as every y is
    }
    for (int j = threadIdx.x; j < nvec; j += kBlock) {
      float l[V], x[V], zh[V], lq[V];
      load_vec<T, V>(lr + (int64_t)j * V, l);
      if (want_soft) {
        logit_x(j, l, x);
#pragma unroll
        for (int i = 0; i < V; ++i) {
          x[i] = to_f32(from_f32<T>(mul_rn(expf(sub_rn(x[i], m2)), rinv)));
          const float h = (x[i] == ymax) ? 1.0f : 0.0f;
          zh[i] = add_rn(to_f32(from_f32<T>(sub_rn(h, x[i]))), x[i]);
        }
        if (y_out) store_vec<T, V>(y_out + row * y_stride + (int64_t)j * V, x);
        if (zh_out) store_vec<T, V>(zh_out + row * zh_stride + (int64_t)j * V, zh);
      }
      if (want_lq) {
#pragma unroll
        for (int i = 0; i < V; ++i) lq[i] = sub_rn(sub_rn(l[i], m1), lse);
        store_vec<T, V>(lq_out + row * (int64_t)K + (int64_t)j * V, lq);
      }
    }
  }
}

template <typename T, int V>
__global__ __launch_bounds__(kBlock) void gumbel_bwd_long_kernel(
    const T* __restrict__ y, int64_t y_stride, const T* __restrict__ lq, const T* __restrict__ gy, int64_t gy_stride,
    const T* __restrict__ gzh, int64_t gzh_stride, const T* __restrict__ glq, T* __restrict__ glogits,
    int64_t gl_stride, int64_t B, int K, float tau) {
  const int nvec = K / V;
  const float itau = 1.0f / tau;
  const bool soft = gy || gzh;
  for (int64_t row = blockIdx.x; row < B; row += gridDim.x) {
    auto upstream = [&](int j, float (&gt)[V]) {   // gy + gzh: the straight-through sample passes its gradient to y
      float q[V];
#pragma unroll
      for (int i = 0; i < V; ++i) gt[i] = 0.f;
      if (gy) load_vec<T, V>(gy + row * gy_stride + (int64_t)j * V, gt);
      if (gzh) {
        load_vec<T, V>(gzh + row * gzh_stride + (int64_t)j * V, q);
#pragma unroll
        for (int i = 0; i < V; ++i) gt[i] += q[i];
      }
    };
    float dot = 0.f, sg = 0.f;
    for (int j = threadIdx.x; j < nvec; j += kBlock) {
      float yv[V], gt[V], gq[V];
      if (soft) {
        load_vec<T, V>(y + row * y_stride + (int64_t)j * V, yv);
        upstream(j, gt);
#pragma unroll
        for (int i = 0; i < V; ++i) dot += gt[i] * yv[i];
      }
      if (glq) {
        load_vec<T, V>(glq + row * (int64_t)K + (int64_t)j * V, gq);
#pragma unroll
        for (int i = 0; i < V; ++i) sg += gq[i];
      }
    }
    if (soft) dot = row_sum(dot, kBlock);
    if (glq) sg = row_sum(sg, kBlock);
    for (int j = threadIdx.x; j < nvec; j += kBlock) {
      float yv[V], gt[V], q[V], gq[V], o[V];
#pragma unroll
      for (int i = 0; i < V; ++i) o[i] = 0.f;
      if (soft) {
        load_vec<T, V>(y + row * y_stride + (int64_t)j * V, yv);
        upstream(j, gt);
#pragma unroll
        for (int i = 0; i < V; ++i) o[i] += itau * yv[i] * (gt[i] - dot);
      }
      if (glq) {
        load_vec<T, V>(lq + row * (int64_t)K + (int64_t)j * V, q);
        load_vec<T, V>(glq + row * (int64_t)K + (int64_t)j * V, gq);
#pragma unroll
        for (int i = 0; i < V; ++i) o[i] += gq[i] - exp_acc(q[i]) * sg;
      }
      store_vec<T, V>(glogits + row * gl_stride + (int64_t)j * V, o);
    }
  }
}

template <typename T, int V, bool BWD>
__global__ __launch_bounds__(kBlock) void categorical_kl_long_kernel(
    const T* __restrict__ a, int64_t stride_a, const T* __restrict__ b, int64_t stride_b, const T* __restrict__ gkl,
    T* __restrict__ kl, T* __restrict__ ga, T* __restrict__ gb, int64_t B, int K) {
  const int nvec = K / V;
  for (int64_t row = blockIdx.x; row < B; row += gridDim.x) {
    const T* ar = a + row * stride_a;
    const T* br = b + row * stride_b;
    float ma = kNegInf, mb = kNegInf;
    for (int j = threadIdx.x; j < nvec; j += kBlock) {
      float va[V], vb[V];
      load_vec<T, V>(ar + (int64_t)j * V, va);
      load_vec<T, V>(br + (int64_t)j * V, vb);
#pragma unroll
      for (int i = 0; i < V; ++i) { ma = fmaxf(ma, va[i]); mb = fmaxf(mb, vb[i]); }
    }
    ma = row_max(ma, kBlock);
    mb = row_max(mb, kBlock);
    float sa = 0.f, sb = 0.f;
    for (int j = threadIdx.x; j < nvec; j += kBlock) {
      float va[V], vb[V];
      load_vec<T, V>(ar + (int64_t)j * V, va);
      load_vec<T, V>(br + (int64_t)j * V, vb);
#pragma unroll
      for (int i = 0; i < V; ++i) { sa += exp_acc(va[i] - ma); sb += exp_acc(vb[i] - mb); }
    }
    sa = row_sum(sa, kBlock);
    sb = row_sum(sb, kBlock);
    const float lsa = ma + logf(sa), lsb = mb + logf(sb);
    const float ia = rcp_nr(sa), ib = rcp_nr(sb);
    float s = 0.f;
    for (int j = threadIdx.x; j < nvec; j += kBlock) {
      float va[V], vb[V];
      load_vec<T, V>(ar + (int64_t)j * V, va);
      load_vec<T, V>(br + (int64_t)j * V, vb);
#pragma unroll
      for (int i = 0; i < V; ++i) s += exp_acc(va[i] - ma) * ia * ((va[i] - lsa) - (vb[i] - lsb));
    }
    s = row_sum(s, kBlock);
    if constexpr (!BWD) {
      if (threadIdx.x == 0) kl[row] = from_f32<T>(s);
    } else {
      const float gk = to_f32(gkl[row]);
      for (int j = threadIdx.x; j < nvec; j += kBlock) {
        float va[V], vb[V], oa[V], ob[V];
        load_vec<T, V>(ar + (int64_t)j * V, va);
        load_vec<T, V>(br + (int64_t)j * V, vb);
#pragma unroll
        for (int i = 0; i < V; ++i) {
          const float p = exp_acc(va[i] - ma) * ia;
          oa[i] = gk * p * (((va[i] - lsa) - (vb[i] - lsb)) - s);
          ob[i] = gk * (exp_acc(vb[i] - mb) * ib - p);
        }
        store_vec<T, V>(ga + row * (int64_t)K + (int64_t)j * V, oa);
        if (gb) store_vec<T, V>(gb + row * (int64_t)K + (int64_t)j * V, ob);
      }
    }
  }
}

constexpr int kLongRowGrid = 8192;   // blocks of a long-row launch (one row each, grid-strided beyond)

// The sampling kernels hold more per chunk than the KL does (noise, the soft sample, the hard mask): four chunks per
// lane cost them occupancy 4 (97-125 registers in fp32, 138-190 in the 16-bit types).  Two chunks on twice the lanes
// measured better wherever a 64-lane group still holds the row (r03, cold-cache, B = 65536: K = 256 all-output forward
// 49.9 -> 47.7 us with in-kernel noise, 61.4 -> 58.5 injected, backward 36.9 -> 35.2; K = 512 forward 100.2 -> 90.5 /
// 117.5 -> 113.3; K = 128 33.6 -> 31.7); the categorical KL keeps four (43.0 vs 44.6 us at K = 512).
static void gumbel_regroup(RowGeom& g, int64_t rows, int64_t vecs) {
  if (knobs().G || knobs().U || g.G >= 64 || g.U != 4) return;
  int G = g.G;
  while (G < 64 && (vecs + G - 1) / G > 2) G <<= 1;
  if ((vecs + G - 1) / G > 2) return;
  regroup(g, rows, G, (vecs + G - 1) / G >= 2 ? 2 : 1, kGridStream);
}

template <typename T>
static int gumbel_fwd_dispatch(const void* logits, int64_t row_stride, const void* unif, void* y, int64_t ys, void* zh,
                               int64_t zs, void* lq,
                               int64_t B, int64_t K, float tau, uint64_t seed, uint64_t offset, int64_t row0,
                               hipStream_t stream) {
  constexpr int VN = VecOf<T>::n;
  const bool vec = (K % VN == 0) && (row_stride % VN == 0) && (ys % VN == 0) && (zs % VN == 0) && aligned16(logits) &&
                   aligned16(y) && aligned16(zh) &&
                   aligned16(lq) && (!unif || aligned16(unif));
  RowGeom g;
  if (!row_geom_resident(B, vec ? K / VN : K, kGridStream, &g)) {
    const int grid = (int)(B < kLongRowGrid ? B : kLongRowGrid);
    if (vec)
      hipLaunchKernelGGL((gumbel_fwd_long_kernel<T, VN>), dim3(grid), dim3(kBlock), 0, stream, (const T*)logits,
                         row_stride, (const T*)unif, (T*)y, ys, (T*)zh, zs, (T*)lq, B, (int)K, tau, seed, offset, row0);
    else
      hipLaunchKernelGGL((gumbel_fwd_long_kernel<T, 1>), dim3(grid), dim3(kBlock), 0, stream, (const T*)logits,
                         row_stride, (const T*)unif, (T*)y, ys, (T*)zh, zs, (T*)lq, B, (int)K, tau, seed, offset, row0);
    return hipGetLastError() == hipSuccess ? VAE_OK : VAE_ERR_LAUNCH;
  }
  if (vec) gumbel_regroup(g, B, K / VN);
#define VAE_GUMBEL_FWD(VV, INJ_)                                                                                   \
  VAE_SWITCH_U(g.U, hipLaunchKernelGGL((gumbel_fwd_kernel<T, VV, U, INJ_>), dim3(g.grid), dim3(kBlock), 0, stream,     \
                                       (const T*)logits, row_stride, (const T*)unif, (T*)y, ys, (T*)zh, zs, (T*)lq, B, \
                                       (int)K, tau, seed, offset, row0, g.G, g.tiles))
  if (vec) {
    if (unif) { VAE_GUMBEL_FWD(VN, true); } else { VAE_GUMBEL_FWD(VN, false); }
  } else {
    if (unif) { VAE_GUMBEL_FWD(1, true); } else { VAE_GUMBEL_FWD(1, false); }
  }
#undef VAE_GUMBEL_FWD
  return hipGetLastError() == hipSuccess ? VAE_OK : VAE_ERR_LAUNCH;
}

template <typename T>
static int gumbel_bwd_dispatch(const void* y, int64_t ys, const void* lq, const void* gy, int64_t gys, const void* gzh,
                               int64_t gzs, const void* glq, void* glogits, int64_t gls, int64_t B, int64_t K, float tau, hipStream_t stream) {
  constexpr int VN = VecOf<T>::n;
  const bool vec = (K % VN == 0) && (ys % VN == 0) && (gys % VN == 0) && (gzs % VN == 0) && (gls % VN == 0) &&
                   aligned16(y) && aligned16(lq) && aligned16(glogits) && (!gy || aligned16(gy)) &&
                   (!gzh || aligned16(gzh)) && (!glq || aligned16(glq));
  RowGeom g;
  if (!row_geom_resident(B, vec ? K / VN : K, kGridStream, &g)) {
    const int grid = (int)(B < kLongRowGrid ? B : kLongRowGrid);
    if (vec)
      hipLaunchKernelGGL((gumbel_bwd_long_kernel<T, VN>), dim3(grid), dim3(kBlock), 0, stream, (const T*)y, ys,
                         (const T*)lq, (const T*)gy, gys, (const T*)gzh, gzs, (const T*)glq, (T*)glogits, gls, B, (int)K,
                         tau);
    else
      hipLaunchKernelGGL((gumbel_bwd_long_kernel<T, 1>), dim3(grid), dim3(kBlock), 0, stream, (const T*)y, ys,
                         (const T*)lq, (const T*)gy, gys, (const T*)gzh, gzs, (const T*)glq, (T*)glogits, gls, B, (int)K,
                         tau);
    return hipGetLastError() == hipSuccess ? VAE_OK : VAE_ERR_LAUNCH;
  }
  if (vec) gumbel_regroup(g, B, K / VN);
  if (vec) {
    VAE_SWITCH_U(g.U, hipLaunchKernelGGL((gumbel_bwd_kernel<T, VN, U>), dim3(g.grid), dim3(kBlock), 0, stream,
                                         (const T*)y, ys, (const T*)lq, (const T*)gy, gys, (const T*)gzh, gzs,
                                         (const T*)glq, (T*)glogits, gls, B, (int)K, tau, g.G, g.tiles));
  } else {
    VAE_SWITCH_U(g.U, hipLaunchKernelGGL((gumbel_bwd_kernel<T, 1, U>), dim3(g.grid), dim3(kBlock), 0, stream,
                                         (const T*)y, ys, (const T*)lq, (const T*)gy, gys, (const T*)gzh, gzs,
                                         (const T*)glq, (T*)glogits, gls, B, (int)K, tau, g.G, g.tiles));
  }
  return hipGetLastError() == hipSuccess ? VAE_OK : VAE_ERR_LAUNCH;
}

template <typename T, bool BWD>
static int cat_kl_dispatch(const void* a, int64_t sa, const void* b, int64_t sb, const void* gkl, void* kl, void* ga,
                           void* gb, int64_t B, int64_t K, hipStream_t stream) {
  constexpr int VN = VecOf<T>::n;
  bool vec = (K % VN == 0) && (sa % VN == 0) && (sb % VN == 0) && aligned16(a) && aligned16(b);
  if (BWD) vec = vec && aligned16(ga) && (!gb || aligned16(gb));
  RowGeom g;
  if (!row_geom_resident(B, vec ? K / VN : K, kGridStream, &g)) {
    const int grid = (int)(B < kLongRowGrid ? B : kLongRowGrid);
    if (vec)
      hipLaunchKernelGGL((categorical_kl_long_kernel<T, VN, BWD>), dim3(grid), dim3(kBlock), 0, stream, (const T*)a, sa,
                         (const T*)b, sb, (const T*)gkl, (T*)kl, (T*)ga, (T*)gb, B, (int)K);
    else
      hipLaunchKernelGGL((categorical_kl_long_kernel<T, 1, BWD>), dim3(grid), dim3(kBlock), 0, stream, (const T*)a, sa,
                         (const T*)b, sb, (const T*)gkl, (T*)kl, (T*)ga, (T*)gb, B, (int)K);
    return hipGetLastError() == hipSuccess ? VAE_OK : VAE_ERR_LAUNCH;
  }
  if (vec) {
    VAE_SWITCH_U(g.U, hipLaunchKernelGGL((categorical_kl_kernel<T, VN, U, BWD>), dim3(g.grid), dim3(kBlock), 0,
                                         stream, (const T*)a, sa, (const T*)b, sb, (const T*)gkl, (T*)kl, (T*)ga,
                                         (T*)gb, B, (int)K, g.G, g.tiles));
  } else {
    VAE_SWITCH_U(g.U, hipLaunchKernelGGL((categorical_kl_kernel<T, 1, U, BWD>), dim3(g.grid), dim3(kBlock), 0, stream,
                                         (const T*)a, sa, (const T*)b, sb, (const T*)gkl, (T*)kl, (T*)ga, (T*)gb, B,
                                         (int)K, g.G, g.tiles));
  }
  return hipGetLastError() == hipSuccess ? VAE_OK : VAE_ERR_LAUNCH;
}

}  // namespace vae

using namespace vae;

extern "C" {

int vae_gumbel_fwd(const void* logits, int64_t row_stride, const void* unif, void* y, int64_t y_stride, void* z_hard,
                   int64_t z_hard_stride, void* log_q_z,
                   int64_t B, int64_t K, float tau, uint64_t seed, uint64_t offset, int64_t row0, int dtype,
                   vae_stream_t stream) {
  if (B < 0 || K <= 0 || K > VAE_MAX_ROW || !(tau > 0.f)) return VAE_ERR_ARG;
  if (B == 0) return VAE_OK;
  if (y_stride == 0) y_stride = K;
  if (z_hard_stride == 0) z_hard_stride = K;
  if (!logits || (!y && !z_hard && !log_q_z) || row_stride < K || y_stride < K || z_hard_stride < K)
    return VAE_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == VAE_F32)
    return gumbel_fwd_dispatch<float>(logits, row_stride, unif, y, y_stride, z_hard, z_hard_stride, log_q_z, B, K, tau, seed, offset, row0, st);
  if (dtype == VAE_BF16)
    return gumbel_fwd_dispatch<bf16_t>(logits, row_stride, unif, y, y_stride, z_hard, z_hard_stride, log_q_z, B, K, tau, seed, offset, row0,
                                       st);
  if (dtype == VAE_F16)
    return gumbel_fwd_dispatch<f16_t>(logits, row_stride, unif, y, y_stride, z_hard, z_hard_stride, log_q_z, B, K, tau, seed, offset, row0,
                                       st);
  return VAE_ERR_ARG;
}

int vae_gumbel_bwd(const void* y, int64_t y_stride, const void* log_q_z, const void* gy, int64_t gy_stride,
                   const void* gz_hard, int64_t gz_hard_stride, const void* glog_q_z, void* glogits,
                   int64_t glogits_stride, int64_t B, int64_t K, float tau, int dtype, vae_stream_t stream) {
  if (B < 0 || K <= 0 || K > VAE_MAX_ROW || !(tau > 0.f)) return VAE_ERR_ARG;
  if (B == 0) return VAE_OK;
  if (y_stride == 0) y_stride = K;
  if (gy_stride == 0) gy_stride = K;
  if (gz_hard_stride == 0) gz_hard_stride = K;
  if (glogits_stride == 0) glogits_stride = K;
  if (!glogits || y_stride < K || gy_stride < K || gz_hard_stride < K || glogits_stride < K) return VAE_ERR_ARG;
  if (((gy || gz_hard) && !y) || (glog_q_z && !log_q_z)) return VAE_ERR_ARG;   // no gradient at all: zeros
  hipStream_t st = (hipStream_t)stream;
  if (dtype == VAE_F32) return gumbel_bwd_dispatch<float>(y, y_stride, log_q_z, gy, gy_stride, gz_hard, gz_hard_stride, glog_q_z, glogits,
                                      glogits_stride, B, K, tau, st);
  if (dtype == VAE_BF16) return gumbel_bwd_dispatch<bf16_t>(y, y_stride, log_q_z, gy, gy_stride, gz_hard, gz_hard_stride, glog_q_z, glogits,
                                       glogits_stride, B, K, tau, st);
  if (dtype == VAE_F16) return gumbel_bwd_dispatch<f16_t>(y, y_stride, log_q_z, gy, gy_stride, gz_hard, gz_hard_stride, glog_q_z, glogits,
                                       glogits_stride, B, K, tau, st);
  return VAE_ERR_ARG;
}

int vae_categorical_kl_fwd(const void* log_q, int64_t stride_q, const void* log_p, int64_t stride_p, void* kl,
                           int64_t B, int64_t K, int dtype, vae_stream_t stream) {
  if (B < 0 || K <= 0 || K > VAE_MAX_ROW) return VAE_ERR_ARG;
  if (B == 0) return VAE_OK;
  if (!log_q || !log_p || !kl || stride_q < K || stride_p < K) return VAE_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == VAE_F32)
    return cat_kl_dispatch<float, false>(log_q, stride_q, log_p, stride_p, nullptr, kl, nullptr, nullptr, B, K, st);
  if (dtype == VAE_BF16)
    return cat_kl_dispatch<bf16_t, false>(log_q, stride_q, log_p, stride_p, nullptr, kl, nullptr, nullptr, B, K, st);
  if (dtype == VAE_F16)
    return cat_kl_dispatch<f16_t, false>(log_q, stride_q, log_p, stride_p, nullptr, kl, nullptr, nullptr, B, K, st);
  return VAE_ERR_ARG;
}

int vae_categorical_kl_bwd(const void* log_q, int64_t stride_q, const void* log_p, int64_t stride_p, const void* gkl,
                           void* g_log_q, void* g_log_p, int64_t B, int64_t K, int dtype, vae_stream_t stream) {
  if (B < 0 || K <= 0 || K > VAE_MAX_ROW) return VAE_ERR_ARG;
  if (B == 0) return VAE_OK;
  if (!log_q || !log_p || !gkl || !g_log_q || stride_q < K || stride_p < K) return VAE_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == VAE_F32)
    return cat_kl_dispatch<float, true>(log_q, stride_q, log_p, stride_p, gkl, nullptr, g_log_q, g_log_p, B, K, st);
  if (dtype == VAE_BF16)
    return cat_kl_dispatch<bf16_t, true>(log_q, stride_q, log_p, stride_p, gkl, nullptr, g_log_q, g_log_p, B, K, st);
  if (dtype == VAE_F16)
    return cat_kl_dispatch<f16_t, true>(log_q, stride_q, log_p, stride_p, gkl, nullptr, g_log_q, g_log_p, B, K, st);
  return VAE_ERR_ARG;
}

}  // extern "C"
