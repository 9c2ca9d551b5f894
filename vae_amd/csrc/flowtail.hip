// Elementwise tails of the flow layers: what is left of a MAF / RealNVP block once the
// stock GEMMs (masked linears, coupling nets) have run.  All HBM-bound row kernels; the running
// `logdets += logdet` of FlowSequential (flows/__init__.py:555-557) is folded into each tail through
// a logdet_in / logdet_out pair, so a block adds no [B]-sized launches of its own.
//
// Replaces, from /root/reference/flows/__init__.py:
//   :121-125, :176-178   MADE / MADESplit 'direct': u = (x - m) * exp(-a); logdet = -sum_f a
//                        (a = tanh(a) first when pre_exp_tanh)
//   :524-528, :458-460   CouplingLayer / AdditiveCouplingLayer 'direct':
//                        log_s = s_raw*(1-mask); t = t_raw*(1-mask); y = x*exp(log_s) + t; logdet = sum log_s
//   :233-257             BatchNormFlow 'direct': batch mean / var (+eps) over the batch, x_hat, y, logdet
//   :392-409             Reverse, when it follows a BatchNormFlow: folded into the store (column F-1-j)
#include "common.hpp"
#include "../../include/vae_amd.h"

namespace vae {

__device__ __forceinline__ float tanh_acc(float v) { return tanhf(v); }

// per-column fp32 vectors (mask, scale, shift, ...): V consecutive floats (V = 1, 4 or 8), cached loads
template <int V>
__device__ __forceinline__ void load_f32(const float* __restrict__ p, float (&out)[V]) {
  if constexpr (V == 1) {
    out[0] = p[0];
  } else {
    typedef float f4 __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int q = 0; q < V / 4; ++q) {
      const f4 r = reinterpret_cast<const f4*>(p)[q];
#pragma unroll
      for (int i = 0; i < 4; ++i) out[4 * q + i] = r[i];
    }
  }
}
template <int V, int U>
__device__ __forceinline__ void load_f32_chunk(const float* __restrict__ base, int j0, int G, int nvec,
                                               float (&out)[U][V]) {
#pragma unroll
  for (int u = 0; u < U; ++u) {
    int j = j0 + u * G;
    j = j < nvec ? j : nvec - 1;
    load_f32<V>(base + (int64_t)j * V, out[u]);
  }
}

// ---------------------------------------------------------------------------------------------
// MADE tail
// ---------------------------------------------------------------------------------------------
template <typename T, int V, int U>
__global__ __launch_bounds__(kBlock) void made_tail_fwd_kernel(
    const T* __restrict__ x, int64_t sx, const T* __restrict__ m, int64_t sm, const T* __restrict__ a, int64_t sa,
    T* __restrict__ u, const T* __restrict__ ld_in, T* __restrict__ ld_out, int64_t B, int F, int pre_tanh, int G,
    int64_t tiles) {
  const int rpb = kBlock / G;
  const bool nts = nt_stores((int64_t)B * F * (int64_t)sizeof(T));
  const int sub = threadIdx.x % G;
  const int rloc = threadIdx.x / G;
  const int nvec = F / V;
  for (int64_t t = blockIdx.x; t < tiles; t += gridDim.x) {
    const int64_t row = t * rpb + rloc;
    float s = 0.f;
    if (row < B) {
      for (int j0 = sub; j0 < nvec; j0 += G * U) {
        float xv[U][V], mv[U][V], av[U][V];
        load_chunk<T, V, U>(x + row * sx, j0, G, nvec, xv);
        load_chunk<T, V, U>(m + row * sm, j0, G, nvec, mv);
        load_chunk<T, V, U>(a + row * sa, j0, G, nvec, av);
#pragma unroll
        for (int k = 0; k < U; ++k) {
          const int j = j0 + k * G;
          if (j < nvec) {
            float o[V];
#pragma unroll
            for (int i = 0; i < V; ++i) {
              const float ai = pre_tanh ? tanh_acc(av[k][i]) : av[k][i];
              o[i] = (xv[k][i] - mv[k][i]) * exp_acc(-ai);
              s -= ai;
            }
            store_vec<T, V>(nts, u + row * F + (int64_t)j * V, o);
          }
        }
      }
    }
    s = group_sum(s, G);
    if (row < B && sub == 0) ld_out[row] = from_f32<T>((ld_in ? to_f32(ld_in[row]) : 0.f) + to_f32(from_f32<T>(s)));
  }
}

// gx = gu*exp(-a); gm = -gx; ga = (-u*gu - gld) [* (1 - tanh(a)^2)]
template <typename T, int V, int U>
__global__ __launch_bounds__(kBlock) void made_tail_bwd_kernel(
    const T* __restrict__ u, const T* __restrict__ a, int64_t sa, const T* __restrict__ gu, const T* __restrict__ gld,
    T* __restrict__ gx, T* __restrict__ gm, int64_t sgm, T* __restrict__ ga, int64_t sga, int64_t B, int F,
    int pre_tanh, int G, int64_t tiles) {
  const int rpb = kBlock / G;
  const bool nts = nt_stores((int64_t)B * F * (int64_t)sizeof(T));
  const int sub = threadIdx.x % G;
  const int rloc = threadIdx.x / G;
  const int nvec = F / V;
  for (int64_t t = blockIdx.x; t < tiles; t += gridDim.x) {
    const int64_t row = t * rpb + rloc;
    if (row >= B) continue;
    const float gl = gld ? to_f32(gld[row]) : 0.f;
    for (int j0 = sub; j0 < nvec; j0 += G * U) {
      float uv[U][V], av[U][V], gv[U][V];
      load_chunk<T, V, U>(u + row * F, j0, G, nvec, uv);
      load_chunk<T, V, U>(a + row * sa, j0, G, nvec, av);
      if (gu) load_chunk<T, V, U>(gu + row * F, j0, G, nvec, gv);
#pragma unroll
      for (int k = 0; k < U; ++k) {
        const int j = j0 + k * G;
        if (j < nvec) {
          float ox[V], om[V], oa[V];
#pragma unroll
          for (int i = 0; i < V; ++i) {
            const float g = gu ? gv[k][i] : 0.f;
            const float ai = pre_tanh ? tanh_acc(av[k][i]) : av[k][i];
            ox[i] = g * exp_acc(-ai);
            om[i] = -ox[i];
            float d = -uv[k][i] * g - gl;
            if (pre_tanh) d *= (1.0f - ai * ai);
            oa[i] = d;
          }
          if (gx) store_vec<T, V>(nts, gx + row * F + (int64_t)j * V, ox);
          store_vec<T, V>(nts, gm + row * sgm + (int64_t)j * V, om);
          store_vec<T, V>(nts, ga + row * sga + (int64_t)j * V, oa);
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// coupling tail (mask: F floats, 1 = pass-through input of the nets, 0 = transformed)
// ---------------------------------------------------------------------------------------------
template <typename T, int V, int U>
__global__ __launch_bounds__(kBlock) void coupling_tail_fwd_kernel(
    const T* __restrict__ x, const T* __restrict__ s_raw, const T* __restrict__ t_raw, const float* __restrict__ mask,
    T* __restrict__ y, const T* __restrict__ ld_in, T* __restrict__ ld_out, int64_t B, int F, int G, int64_t tiles) {
  const int rpb = kBlock / G;
  const bool nts = nt_stores((int64_t)B * F * (int64_t)sizeof(T));
  const int sub = threadIdx.x % G;
  const int rloc = threadIdx.x / G;
  const int nvec = F / V;
  for (int64_t t = blockIdx.x; t < tiles; t += gridDim.x) {
    const int64_t row = t * rpb + rloc;
    float s = 0.f;
    if (row < B) {
      for (int j0 = sub; j0 < nvec; j0 += G * U) {
        float xv[U][V], sv[U][V], tv[U][V], mk[U][V];
        load_chunk<T, V, U>(x + row * F, j0, G, nvec, xv);
        if (s_raw) load_chunk<T, V, U>(s_raw + row * F, j0, G, nvec, sv);
        load_chunk<T, V, U>(t_raw + row * F, j0, G, nvec, tv);
        load_f32_chunk<V, U>(mask, j0, G, nvec, mk);
#pragma unroll
        for (int k = 0; k < U; ++k) {
          const int j = j0 + k * G;
          if (j < nvec) {
            float o[V];
#pragma unroll
            for (int i = 0; i < V; ++i) {
              const float inv = 1.0f - mk[k][i];
              const float ls = s_raw ? mul_rn(sv[k][i], inv) : 0.f;
              // separate roundings, as the reference's x * exp(log_s) + t (flows/__init__.py:524-528): no FMA
              // contraction, so the vector and scalar instantiations agree bit for bit
              o[i] = add_rn(s_raw ? mul_rn(xv[k][i], exp_acc(ls)) : xv[k][i], mul_rn(tv[k][i], inv));
              s += ls;
            }
            store_vec<T, V>(nts, y + row * F + (int64_t)j * V, o);
          }
        }
      }
    }
    s = group_sum(s, G);
    if (row < B && sub == 0) ld_out[row] = from_f32<T>((ld_in ? to_f32(ld_in[row]) : 0.f) + to_f32(from_f32<T>(s)));
  }
}

// gx = gy*exp(log_s) (direct path only); gs_raw = (gy*x*exp(log_s) + gld)*(1-mask); gt_raw = gy*(1-mask)
template <typename T, int V, int U>
__global__ __launch_bounds__(kBlock) void coupling_tail_bwd_kernel(
    const T* __restrict__ x, const T* __restrict__ s_raw, const float* __restrict__ mask, const T* __restrict__ gy,
    const T* __restrict__ gld, T* __restrict__ gx, T* __restrict__ gs, T* __restrict__ gt, int64_t B, int F, int G,
    int64_t tiles) {
  const int rpb = kBlock / G;
  const bool nts = nt_stores((int64_t)B * F * (int64_t)sizeof(T));
  const int sub = threadIdx.x % G;
  const int rloc = threadIdx.x / G;
  const int nvec = F / V;
  for (int64_t t = blockIdx.x; t < tiles; t += gridDim.x) {
    const int64_t row = t * rpb + rloc;
    if (row >= B) continue;
    const float gl = gld ? to_f32(gld[row]) : 0.f;
    for (int j0 = sub; j0 < nvec; j0 += G * U) {
      float xv[U][V], sv[U][V], gv[U][V], mk[U][V];
      if (s_raw) {
        load_chunk<T, V, U>(x + row * F, j0, G, nvec, xv);
        load_chunk<T, V, U>(s_raw + row * F, j0, G, nvec, sv);
      }
      if (gy) load_chunk<T, V, U>(gy + row * F, j0, G, nvec, gv);
      load_f32_chunk<V, U>(mask, j0, G, nvec, mk);
#pragma unroll
      for (int k = 0; k < U; ++k) {
        const int j = j0 + k * G;
        if (j < nvec) {
          float ox[V], os[V], ot[V];
#pragma unroll
          for (int i = 0; i < V; ++i) {
            const float g = gy ? gv[k][i] : 0.f;
            const float inv = 1.0f - mk[k][i];
            const float e = s_raw ? exp_acc(sv[k][i] * inv) : 1.0f;
            ox[i] = g * e;
            os[i] = s_raw ? (g * xv[k][i] * e + gl) * inv : 0.f;
            ot[i] = g * inv;
          }
          const int64_t o = row * F + (int64_t)j * V;
          if (gx) store_vec<T, V>(nts, gx + o, ox);
          if (gs) store_vec<T, V>(nts, gs + o, os);
          store_vec<T, V>(nts, gt + o, ot);
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// BatchNormFlow: column moments over the batch.
//
// A lane group owns a row, a lane always the same columns (sub + k*G), so every thread keeps the
// two running sums of ITS columns in registers while the block walks its row tiles; one LDS step
// folds the rows of a block, and each block writes one [2][F] partial (plus one float, the
// block's sum of gld) that a small second kernel adds up in block order, in double.
//   mode 0 (forward):  f1 = x - c, f2 = (x - c)^2, c = x[0, :]  (shifted: no cancellation)
//   mode 1 (backward): f1 = gy,    f2 = gy * x_hat;  extra = sum_b gld[b]
// `rev`: gy / y live in reversed column order (the fused Reverse).
// ---------------------------------------------------------------------------------------------
constexpr int kBnChunks = 4;                    // vectors per lane: F <= 4 * G * V
constexpr int kBnGrid = 1024;                   // 4 blocks per CU; partials = 1024 * (2F + 1) floats

template <typename T, int V>
__device__ __forceinline__ void load_vec_rev(const T* __restrict__ rowp, int j, int nvec, bool rev, float (&out)[V]) {
  if (!rev) {
    load_vec<T, V>(rowp + (int64_t)j * V, out);
  } else {
    float tmp[V];
    load_vec<T, V>(rowp + (int64_t)(nvec - 1 - j) * V, tmp);
#pragma unroll
    for (int i = 0; i < V; ++i) out[i] = tmp[V - 1 - i];
  }
}

// NJ column-vectors per lane (1, 2 or 4), R row tiles in flight per iteration (all loads issued up front).
template <typename T, int V, int NJ, int R>
__global__ __launch_bounds__(kBlock) void bn_colstats_kernel(
    const T* __restrict__ x, const T* __restrict__ gy, const T* __restrict__ gld, const float* __restrict__ mean,
    const float* __restrict__ rstd, float* __restrict__ part, int64_t B, int F, int mode, int rev, int G,
    int64_t tiles, int jbase, int wvec) {
  // columns: the window of `wvec` vectors that starts at vector `jbase` (the whole row when it fits kBnChunks
  // vectors per lane; rows wider than that take one launch per window of 4 * kBlock vectors)
  __shared__ float lds[NJ * V * kBlock];   // [slot][tid], slot < NJ*V; one statistic at a time
  const int rpb = kBlock / G;
  const int sub = threadIdx.x % G;
  const int rloc = threadIdx.x / G;
  const int nvec = F / V;
  float a1[NJ][V], a2[NJ][V], c0[NJ][V], c1[NJ][V];
  int jc[NJ];
  bool jv[NJ];
  float extra = 0.f;
#pragma unroll
  for (int k = 0; k < NJ; ++k) {
    const int j = sub + k * G;
    jv[k] = j < wvec;
    jc[k] = jbase + (jv[k] ? j : wvec - 1);
#pragma unroll
    for (int i = 0; i < V; ++i) { a1[k][i] = 0.f; a2[k][i] = 0.f; c1[k][i] = 1.f; }
    if (mode == 0) {
      load_vec<T, V, false>(x + (int64_t)jc[k] * V, c0[k]);          // shift: the first row
    } else {
      load_f32<V>(mean + (int64_t)jc[k] * V, c0[k]);
      load_f32<V>(rstd + (int64_t)jc[k] * V, c1[k]);
    }
  }
  for (int64_t t0 = blockIdx.x; t0 < tiles; t0 += (int64_t)gridDim.x * R) {
    float xv[R][NJ][V], gv[R][NJ][V];
    bool rv[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int64_t t = t0 + (int64_t)r * gridDim.x;
      const int64_t row = t * rpb + rloc;
      rv[r] = t < tiles && row < B;
      const int64_t rowc = rv[r] ? row : 0;
#pragma unroll
      for (int k = 0; k < NJ; ++k) {
        load_vec<T, V>(x + rowc * F + (int64_t)jc[k] * V, xv[r][k]);
        if (mode == 1) load_vec_rev<T, V>(gy + rowc * F, jc[k], nvec, rev != 0, gv[r][k]);
      }
      if (mode == 1 && gld && sub == 0 && jbase == 0 && rv[r]) extra += to_f32(gld[row]);
    }
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int k = 0; k < NJ; ++k) {
        const bool ok = rv[r] && jv[k];
#pragma unroll
        for (int i = 0; i < V; ++i) {
          if (mode == 0) {
            const float d = ok ? xv[r][k][i] - c0[k][i] : 0.f;
            a1[k][i] += d;
            a2[k][i] += d * d;
          } else {
            const float g = ok ? gv[r][k][i] : 0.f;
            a1[k][i] += g;
            a2[k][i] += g * ((xv[r][k][i] - c0[k][i]) * c1[k][i]);
          }
        }
      }
  }
  // fold the rpb rows of the block (slot-major LDS: conflict-free), then one partial per block
  float* mypart = part + (size_t)blockIdx.x * (2 * (size_t)F + 1);
  if (rpb > 1) {
    for (int st = 0; st < 2; ++st) {
      if (st) __syncthreads();   // the readers of the first statistic are done
#pragma unroll
      for (int k = 0; k < NJ; ++k)
#pragma unroll
        for (int i = 0; i < V; ++i) lds[(k * V + i) * kBlock + threadIdx.x] = st ? a2[k][i] : a1[k][i];
      __syncthreads();
      // thread (rloc, sub) folds the slots q = rloc, rloc + rpb, ... of lane `sub`
      for (int q = rloc; q < NJ * V; q += rpb) {
        const int k = q / V, i = q % V;
        const int j = sub + k * G;
        if (j < wvec) {
          float s = 0.f;
          for (int r = 0; r < rpb; ++r) s += lds[q * kBlock + r * G + sub];
          mypart[(size_t)st * F + (size_t)(jbase + j) * V + i] = s;
        }
      }
    }
  } else {
#pragma unroll
    for (int k = 0; k < NJ; ++k) {
      if (jv[k]) {
#pragma unroll
        for (int i = 0; i < V; ++i) {
          mypart[(size_t)jc[k] * V + i] = a1[k][i];
          mypart[(size_t)F + (size_t)jc[k] * V + i] = a2[k][i];
        }
      }
    }
  }
  // the block's sum of gld: lanes with sub == 0 hold it
  __shared__ float ex[kBlock / 64];
  float e = wave_sum(extra);
  if ((threadIdx.x & 63) == 0) ex[threadIdx.x >> 6] = e;
  __syncthreads();
  if (threadIdx.x == 0 && jbase == 0) mypart[2 * (size_t)F] = ex[0] + ex[1] + ex[2] + ex[3];
}

// MADE tail AND the column moments of its output, for the BatchNormFlow that follows it in every MAF block
// (flows/__init__.py:582-593: MADE, BatchNormFlow, Reverse): u is written once and its moments are taken from the
// registers that hold it, so the BatchNormFlow forward no longer reads u a first time for its statistics (a MAF block
// forward moves 24F + 12 bytes per row instead of 28F + 12).  Geometry and partial layout are bn_colstats_kernel's
// (mode 0: shifted sums, the shift being u[0, :], which every lane forms for its own columns from row 0).
// The values entering the sums are u as STORED (rounded to the tensor dtype): what BatchNormFlow would read back.
template <typename T, int V, int NJ, int R>
__global__ __launch_bounds__(kBlock) void made_tail_moments_fwd_kernel(
    const T* __restrict__ x, int64_t sx, const T* __restrict__ m, int64_t sm, const T* __restrict__ a, int64_t sa,
    T* __restrict__ u, const T* __restrict__ ld_in, T* __restrict__ ld_out, float* __restrict__ part, int64_t B, int F,
    int pre_tanh, int G, int64_t tiles) {
  __shared__ float lds[NJ * V * kBlock];
  __shared__ float wsum[kBlock / 64];
  const int rpb = kBlock / G;
  const bool nts = nt_stores((int64_t)B * F * (int64_t)sizeof(T));
  const int sub = threadIdx.x % G;
  const int rloc = threadIdx.x / G;
  const int nvec = F / V;
  float a1[NJ][V], a2[NJ][V], c0[NJ][V];
  int jc[NJ];
  bool jv[NJ];
  auto tail = [&](const float (&xv)[V], const float (&mv)[V], const float (&av)[V], float (&o)[V], float& s) {
#pragma unroll
    for (int i = 0; i < V; ++i) {
      const float ai = pre_tanh ? tanh_acc(av[i]) : av[i];
      o[i] = (xv[i] - mv[i]) * exp_acc(-ai);
      s -= ai;
    }
  };
#pragma unroll
  for (int k = 0; k < NJ; ++k) {
    const int j = sub + k * G;
    jv[k] = j < nvec;
    jc[k] = jv[k] ? j : nvec - 1;
    float xv[V], mv[V], av[V], o[V], dummy = 0.f;
    load_vec<T, V, false>(x + (int64_t)jc[k] * V, xv);
    load_vec<T, V, false>(m + (int64_t)jc[k] * V, mv);
    load_vec<T, V, false>(a + (int64_t)jc[k] * V, av);
    tail(xv, mv, av, o, dummy);
#pragma unroll
    for (int i = 0; i < V; ++i) {
      c0[k][i] = to_f32(from_f32<T>(o[i]));
      a1[k][i] = 0.f;
      a2[k][i] = 0.f;
    }
  }
  for (int64_t t0 = blockIdx.x; t0 < tiles; t0 += (int64_t)gridDim.x * R) {
    float xv[R][NJ][V], mv[R][NJ][V], av[R][NJ][V];
    bool rv[R];
    int64_t rows[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int64_t t = t0 + (int64_t)r * gridDim.x;
      const int64_t row = t * rpb + rloc;
      rv[r] = t < tiles && row < B;
      rows[r] = rv[r] ? row : 0;
#pragma unroll
      for (int k = 0; k < NJ; ++k) {
        load_vec<T, V>(x + rows[r] * sx + (int64_t)jc[k] * V, xv[r][k]);
        load_vec<T, V>(m + rows[r] * sm + (int64_t)jc[k] * V, mv[r][k]);
        load_vec<T, V>(a + rows[r] * sa + (int64_t)jc[k] * V, av[r][k]);
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < NJ; ++k) {
        float o[V], sk = 0.f;
        tail(xv[r][k], mv[r][k], av[r][k], o, sk);
        const bool ok = rv[r] && jv[k];
        s += ok ? sk : 0.f;
#pragma unroll
        for (int i = 0; i < V; ++i) {
          const float d = ok ? to_f32(from_f32<T>(o[i])) - c0[k][i] : 0.f;
          a1[k][i] += d;
          a2[k][i] += d * d;
        }
        if (ok) store_vec<T, V>(nts, u + rows[r] * F + (int64_t)jc[k] * V, o);
      }
      // every thread of the block reaches this (uniform trip counts); bn_geom's G may be 128: two rows a block
      if (G <= 64) {
        s = group_sum(s, G);
      } else {
        s = group_sum(s, 64);
        lds_barrier();
        if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = s;
        lds_barrier();
        s = 0.f;
        for (int w = 0; w < G / 64; ++w) s += wsum[rloc * (G / 64) + w];
      }
      if (rv[r] && sub == 0)
        ld_out[rows[r]] = from_f32<T>((ld_in ? to_f32(ld_in[rows[r]]) : 0.f) + to_f32(from_f32<T>(s)));
    }
  }
  // fold the rpb rows of the block, one partial per block: bn_colstats_kernel's layout
  float* mypart = part + (size_t)blockIdx.x * (2 * (size_t)F + 1);
  if (rpb > 1) {
    for (int st = 0; st < 2; ++st) {
      __syncthreads();
#pragma unroll
      for (int k = 0; k < NJ; ++k)
#pragma unroll
        for (int i = 0; i < V; ++i) lds[(k * V + i) * kBlock + threadIdx.x] = st ? a2[k][i] : a1[k][i];
      __syncthreads();
      for (int q = rloc; q < NJ * V; q += rpb) {
        const int k = q / V, i = q % V;
        const int j = sub + k * G;
        if (j < nvec) {
          float sum = 0.f;
          for (int r = 0; r < rpb; ++r) sum += lds[q * kBlock + r * G + sub];
          mypart[(size_t)st * F + (size_t)j * V + i] = sum;
        }
      }
    }
  } else {
#pragma unroll
    for (int k = 0; k < NJ; ++k) {
      if (jv[k]) {
#pragma unroll
        for (int i = 0; i < V; ++i) {
          mypart[(size_t)jc[k] * V + i] = a1[k][i];
          mypart[(size_t)F + (size_t)jc[k] * V + i] = a2[k][i];
        }
      }
    }
  }
  if (threadIdx.x == 0) mypart[2 * (size_t)F] = 0.f;
}

// Second step: 8 columns x 32 partial-groups per block; block order, double.
//   mode 0: mean = c + S1/B, var = S2/B - (S1/B)^2 + eps; then the per-column constants of the apply:
//           rstd, scale = exp(lg) * rstd, beta, ldc = lg - 0.5 log var
//   mode 1: S1 = sum gy (= d beta), S2 = sum gy*x_hat, GLD = sum gld, and the backward apply's constants
//           k1 = scale*S1/B, K = rstd*(scale*S2/B + GLD*rstd/B)   (training; eval: both 0)
constexpr int kFinCols = 8, kFinGroups = kBlock / kFinCols;
__global__ __launch_bounds__(kBlock) void bn_finalize_kernel(
    const float* __restrict__ part, int nparts, const void* __restrict__ x0, int x_dtype, const float* __restrict__ lg,
    const float* __restrict__ beta, float* __restrict__ stats, int64_t B, int F, float eps, int mode, int training) {
  __shared__ double l1[kFinGroups][kFinCols], l2[kFinGroups][kFinCols];
  __shared__ double lex[kBlock / 64];
  const int cl = threadIdx.x % kFinCols, pg = threadIdx.x / kFinCols;
  const int col = blockIdx.x * kFinCols + cl;
  const size_t pstride = 2 * (size_t)F + 1;
  float *o_mean = stats, *o_var = stats + F, *o_rstd = stats + 2 * (size_t)F, *o_scale = stats + 3 * (size_t)F,
        *o_beta = stats + 4 * (size_t)F, *o_ldc = stats + 5 * (size_t)F, *o_s1 = stats + 6 * (size_t)F,
        *o_s2 = stats + 7 * (size_t)F, *o_gsum = stats + 8 * (size_t)F, *o_k1 = stats + 8 * (size_t)F + 4,
        *o_K = stats + 9 * (size_t)F + 4;
  double s1 = 0.0, s2 = 0.0;
  if (col < F) {
#pragma unroll 4
    for (int p = pg; p < nparts; p += kFinGroups) {
      s1 += (double)part[p * pstride + col];
      s2 += (double)part[p * pstride + F + col];
    }
  }
  l1[pg][cl] = s1;
  l2[pg][cl] = s2;
  double gsum = 0.0;
  if (mode == 1) {   // every block needs GLD: thread-strided, then wave + LDS fold (fixed order)
    double e = 0.0;
    for (int p = threadIdx.x; p < nparts; p += kBlock) e += (double)part[p * pstride + 2 * (size_t)F];
    for (int o = 32; o > 0; o >>= 1) e += __shfl_xor(e, o, 64);
    if ((threadIdx.x & 63) == 0) lex[threadIdx.x >> 6] = e;
  }
  __syncthreads();
  if (mode == 1) gsum = lex[0] + lex[1] + lex[2] + lex[3];
  if (pg == 0 && col < F) {
    s1 = 0.0; s2 = 0.0;
    for (int g = 0; g < kFinGroups; ++g) { s1 += l1[g][cl]; s2 += l2[g][cl]; }
    if (mode == 0) {
      const float c = x_dtype == 1 ? to_f32(((const bf16_t*)x0)[col])
                                   : x_dtype == 2 ? to_f32(((const f16_t*)x0)[col]) : ((const float*)x0)[col];
      const double m1 = s1 / (double)B;
      const float mean = (float)((double)c + m1);
      const float var = (float)(s2 / (double)B - m1 * m1) + eps;
      const float r = 1.0f / sqrtf(var);
      o_mean[col] = mean;
      o_var[col] = var;
      o_rstd[col] = r;
      o_scale[col] = expf(lg[col]) * r;
      o_beta[col] = beta[col];
      o_ldc[col] = lg[col] - 0.5f * logf(var);
    } else {
      o_s1[col] = (float)s1;
      o_s2[col] = (float)s2;
      const double inv_b = 1.0 / (double)B;
      const double sc = (double)o_scale[col], r = (double)o_rstd[col];
      o_k1[col] = training ? (float)(sc * s1 * inv_b) : 0.f;
      o_K[col] = training ? (float)(r * (sc * s2 * inv_b + gsum * inv_b * r)) : 0.f;
    }
  }
  if (mode == 1 && blockIdx.x == 0 && threadIdx.x == 0) o_gsum[0] = (float)gsum;
}

// Backward constants from column sums that the caller has reduced over every rank's rows (stats[6F, 8F]):
// k1 = scale*S1/N, K = rstd*(scale*S2/N + GLD*rstd/N) with N the number of rows the statistics were taken over.
__global__ __launch_bounds__(kBlock) void bn_bwd_consts_kernel(float* __restrict__ stats, int F, int64_t count) {
  const int col = blockIdx.x * kBlock + threadIdx.x;
  if (col >= F) return;
  const double inv_n = 1.0 / (double)count;
  const double sc = (double)stats[3 * (size_t)F + col], r = (double)stats[2 * (size_t)F + col];
  const double s1 = (double)stats[6 * (size_t)F + col], s2 = (double)stats[7 * (size_t)F + col];
  const double gsum = (double)stats[8 * (size_t)F];
  stats[8 * (size_t)F + 4 + col] = (float)(sc * s1 * inv_n);
  stats[9 * (size_t)F + 4 + col] = (float)(r * (sc * s2 * inv_n + gsum * inv_n * r));
}

// eval mode / explicit statistics: the per-column constants from given mean / var vectors
__global__ __launch_bounds__(kBlock) void bn_prepare_kernel(const float* __restrict__ mean, const float* __restrict__ var,
                                                           const float* __restrict__ lg, const float* __restrict__ beta,
                                                           float* __restrict__ stats, int F) {
  const int col = blockIdx.x * kBlock + threadIdx.x;
  if (col >= F) return;
  const float r = 1.0f / sqrtf(var[col]);
  stats[col] = mean[col];
  stats[(size_t)F + col] = var[col];
  stats[2 * (size_t)F + col] = r;
  stats[3 * (size_t)F + col] = expf(lg[col]) * r;
  stats[4 * (size_t)F + col] = beta[col];
  stats[5 * (size_t)F + col] = lg[col] - 0.5f * logf(var[col]);
}

// sum of F floats, the same value (same order) in every thread of every block
__device__ __forceinline__ float block_sum_vector(const float* __restrict__ v, int F) {
  __shared__ float bs[kBlock / 64];
  float s = 0.f;
  for (int j = threadIdx.x; j < F; j += kBlock) s += v[j];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) bs[threadIdx.x >> 6] = s;
  __syncthreads();
  return (bs[0] + bs[1]) + (bs[2] + bs[3]);
}

// y[b, p(j)] = (x - mean_j)*scale_j + beta_j (the subtraction first, as the reference: x*scale + shift would
// cancel catastrophically when |mean| >> std);  logdet_out[b] = logdet_in[b] + sum_j ldc_j.
// HOIST (the row fits U vectors per lane): a lane always owns the same columns, so their constants are loaded
// once and stay in registers while the block walks its row tiles.
template <typename T, int V, int U, bool HOIST>
__global__ __launch_bounds__(kBlock) void bn_apply_fwd_kernel(
    const T* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ scale,
    const float* __restrict__ shift, const float* __restrict__ ldc, T* __restrict__ y, const T* __restrict__ ld_in,
    T* __restrict__ ld_out, int64_t B, int F, int rev, int G, int64_t tiles) {
  const int rpb = kBlock / G;
  const bool nts = nt_stores((int64_t)B * F * (int64_t)sizeof(T));
  const int sub = threadIdx.x % G;
  const int rloc = threadIdx.x / G;
  const int nvec = F / V;
  const float ldsum = to_f32(from_f32<T>(block_sum_vector(ldc, F)));
  float mu[U][V], sc[U][V], sh[U][V];
  if constexpr (HOIST) {
    load_f32_chunk<V, U>(mean, sub, G, nvec, mu);
    load_f32_chunk<V, U>(scale, sub, G, nvec, sc);
    load_f32_chunk<V, U>(shift, sub, G, nvec, sh);
  }
  for (int64_t t = blockIdx.x; t < tiles; t += gridDim.x) {
    const int64_t row = t * rpb + rloc;
    if (row >= B) continue;
    for (int j0 = sub; j0 < nvec; j0 += G * U) {
      float xv[U][V];
      load_chunk<T, V, U>(x + row * F, j0, G, nvec, xv);
      if constexpr (!HOIST) {
        load_f32_chunk<V, U>(mean, j0, G, nvec, mu);
        load_f32_chunk<V, U>(scale, j0, G, nvec, sc);
        load_f32_chunk<V, U>(shift, j0, G, nvec, sh);
      }
#pragma unroll
      for (int k = 0; k < U; ++k) {
        const int j = j0 + k * G;
        if (j < nvec) {
          float o[V];
#pragma unroll
          for (int i = 0; i < V; ++i) {
            const float v = (xv[k][i] - mu[k][i]) * sc[k][i] + sh[k][i];
            if (rev) o[V - 1 - i] = v; else o[i] = v;
          }
          store_vec<T, V>(nts, y + row * F + (int64_t)(rev ? nvec - 1 - j : j) * V, o);
        }
      }
    }
    if (sub == 0) ld_out[row] = from_f32<T>((ld_in ? to_f32(ld_in[row]) : 0.f) + ldsum);
  }
}

// gx = scale*gy - k1 - (x - mean)*K   (training: k1, K from the finalize step; eval: k1 = K = 0)
template <typename T, int V, int U, bool HOIST>
__global__ __launch_bounds__(kBlock) void bn_apply_bwd_kernel(
    const T* __restrict__ x, const T* __restrict__ gy, const float* __restrict__ mean, const float* __restrict__ scale,
    const float* __restrict__ k1, const float* __restrict__ kk, T* __restrict__ gx, int64_t B, int F, int training,
    int rev, int G, int64_t tiles) {
  const int rpb = kBlock / G;
  const bool nts = nt_stores((int64_t)B * F * (int64_t)sizeof(T));
  const int sub = threadIdx.x % G;
  const int rloc = threadIdx.x / G;
  const int nvec = F / V;
  float mu[U][V], sc[U][V], c1[U][V], c2[U][V];
  if constexpr (HOIST) {
    load_f32_chunk<V, U>(mean, sub, G, nvec, mu);
    load_f32_chunk<V, U>(scale, sub, G, nvec, sc);
    load_f32_chunk<V, U>(k1, sub, G, nvec, c1);
    load_f32_chunk<V, U>(kk, sub, G, nvec, c2);
  }
  for (int64_t t = blockIdx.x; t < tiles; t += gridDim.x) {
    const int64_t row = t * rpb + rloc;
    if (row >= B) continue;
    for (int j0 = sub; j0 < nvec; j0 += G * U) {
      float xv[U][V], gv[U][V];
      if (training) load_chunk<T, V, U>(x + row * F, j0, G, nvec, xv);
#pragma unroll
      for (int k = 0; k < U; ++k) {
        int jj = j0 + k * G;
        jj = jj < nvec ? jj : nvec - 1;
        load_vec_rev<T, V>(gy + row * F, jj, nvec, rev != 0, gv[k]);
      }
      if constexpr (!HOIST) {
        load_f32_chunk<V, U>(mean, j0, G, nvec, mu);
        load_f32_chunk<V, U>(scale, j0, G, nvec, sc);
        load_f32_chunk<V, U>(k1, j0, G, nvec, c1);
        load_f32_chunk<V, U>(kk, j0, G, nvec, c2);
      }
#pragma unroll
      for (int k = 0; k < U; ++k) {
        const int j = j0 + k * G;
        if (j < nvec) {
          float o[V];
#pragma unroll
          for (int i = 0; i < V; ++i)
            o[i] = training ? sc[k][i] * gv[k][i] - c1[k][i] - (xv[k][i] - mu[k][i]) * c2[k][i] : sc[k][i] * gv[k][i];
          store_vec<T, V>(nts, gx + row * F + (int64_t)j * V, o);
        }
      }
    }
  }
}

// geometry for the column-moment kernel: >= 2 vectors per lane up to a whole wave per row, then wider groups
// until the row fits kBnChunks vectors per lane
static void bn_geom(int64_t B, int64_t nvec, int* G_out, int* NJ_out, int64_t* tiles, int* grid) {
  int G = 1;
  while (G < 64 && (int64_t)G * 2 < nvec) G <<= 1;
  while (G < kBlock && (int64_t)G * kBnChunks < nvec) G <<= 1;
  // wider than kBlock * kBnChunks vectors: the block owns a row and the launch is repeated per column window
  const int64_t per = (int64_t)G * kBnChunks < nvec ? kBnChunks : (nvec + G - 1) / G;
  *NJ_out = per >= 3 ? 4 : (per == 2 ? 2 : 1);
  const int rpb = kBlock / G;
  *G_out = G;
  *tiles = (B + rpb - 1) / rpb;
  *grid = (int)(*tiles < kBnGrid ? *tiles : kBnGrid);
}

template <typename T, int V>
static void bn_colstats_launch(int NJ, int grid, hipStream_t st, const T* x, const T* gy, const T* gld, const float* mean,
                               const float* rstd, float* part, int64_t B, int F, int mode, int rev, int G,
                               int64_t tiles) {
  const int nvec = F / V;
  // row tiles in flight: 8 column-vectors of loads per lane and tensor for 4-element vectors, 4 for the 8-element
  // vectors of the 16-bit types (their registers hold twice the floats: NJ * R = 8 spilled 64-112 bytes per lane to
  // scratch there, -Rpass-analysis=kernel-resource-usage, r03)
  constexpr int R2 = V >= 8 ? 2 : 4;
  // four 8-element vectors per lane do not fit either (the compiler pairs the 32 shift constants for packed
  // arithmetic and parks them in scratch): wide 16-bit rows take windows of two vectors per lane instead
  constexpr int NJW = V >= 8 ? 2 : 4;
  if (NJ == 1)
    hipLaunchKernelGGL((bn_colstats_kernel<T, V, 1, 4>), dim3(grid), dim3(kBlock), 0, st, x, gy, gld, mean, rstd, part,
                       B, F, mode, rev, G, tiles, 0, nvec);
  else if (NJ == 2)
    hipLaunchKernelGGL((bn_colstats_kernel<T, V, 2, R2>), dim3(grid), dim3(kBlock), 0, st, x, gy, gld, mean, rstd, part,
                       B, F, mode, rev, G, tiles, 0, nvec);
  else
    for (int j0 = 0; j0 < nvec; j0 += G * NJW) {   // one window unless the row is wider than G * NJW vectors
      const int w = nvec - j0 < G * NJW ? nvec - j0 : G * NJW;
      hipLaunchKernelGGL((bn_colstats_kernel<T, V, NJW, (V >= 8 ? R2 : 2)>), dim3(grid), dim3(kBlock), 0, st, x, gy, gld,
                         mean, rstd, part, B, F, mode, rev, G, tiles, j0, w);
    }
}

template <typename T>
static int made_fwd_dispatch(const void* x, int64_t sx, const void* m, int64_t sm, const void* a, int64_t sa, void* u,
                             const void* ld_in, void* ld_out, int64_t B, int64_t F, int pre_tanh, hipStream_t st) {
  constexpr int VN = VecOf<T>::n;
  const bool vec = (F % VN == 0) && (sx % VN == 0) && (sm % VN == 0) && (sa % VN == 0) && aligned16(x) &&
                   aligned16(m) && aligned16(a) && aligned16(u);
  const RowGeom g = row_geom(B, vec ? F / VN : F, kGridStream);
  if (vec) {
    VAE_SWITCH_U(g.U, hipLaunchKernelGGL((made_tail_fwd_kernel<T, VN, U>), dim3(g.grid), dim3(kBlock), 0, st,
                                         (const T*)x, sx, (const T*)m, sm, (const T*)a, sa, (T*)u, (const T*)ld_in,
                                         (T*)ld_out, B, (int)F, pre_tanh, g.G, g.tiles));
  } else {
    VAE_SWITCH_U(g.U, hipLaunchKernelGGL((made_tail_fwd_kernel<T, 1, U>), dim3(g.grid), dim3(kBlock), 0, st,
                                         (const T*)x, sx, (const T*)m, sm, (const T*)a, sa, (T*)u, (const T*)ld_in,
                                         (T*)ld_out, B, (int)F, pre_tanh, g.G, g.tiles));
  }
  return hipGetLastError() == hipSuccess ? VAE_OK : VAE_ERR_LAUNCH;
}

// the MADE tail that also leaves the column moments of u in the following BatchNormFlow's stats buffer
template <typename T>
static int made_moments_dispatch(const void* x, int64_t sx, const void* m, int64_t sm, const void* a, int64_t sa, void* u,
                                 const void* ld_in, void* ld_out, float* stats, int64_t B, int64_t F, int pre_tanh,
                                 hipStream_t st) {
  constexpr int VN = VecOf<T>::n;
  const bool vec = (F % VN == 0) && (sx % VN == 0) && (sm % VN == 0) && (sa % VN == 0) && aligned16(x) &&
                   aligned16(m) && aligned16(a) && aligned16(u) && aligned16(stats);
  if (!vec) return VAE_ERR_ARG;
  int G, NJ, grid;
  int64_t tiles;
  bn_geom(B, F / VN, &G, &NJ, &tiles, &grid);
  if ((int64_t)G * kBnChunks < F / VN) return VAE_ERR_ARG;   // wider than one column window
  float* part = stats + 10 * F + 4;
#define VAE_MADE_MOM(NJ_, R_)                                                                                         \
  hipLaunchKernelGGL((made_tail_moments_fwd_kernel<T, VN, NJ_, R_>), dim3(grid), dim3(kBlock), 0, st, (const T*)x, sx, \
                     (const T*)m, sm, (const T*)a, sa, (T*)u, (const T*)ld_in, (T*)ld_out, part, B, (int)F, pre_tanh, G, \
                     tiles)
  if (NJ == 1) { VAE_MADE_MOM(1, 2); } else if (NJ == 2) { VAE_MADE_MOM(2, 2); } else { VAE_MADE_MOM(4, 1); }
#undef VAE_MADE_MOM
  return hipGetLastError() == hipSuccess ? VAE_OK : VAE_ERR_LAUNCH;
}

template <typename T>
static int made_bwd_dispatch(const void* u, const void* a, int64_t sa, const void* gu, const void* gld, void* gx,
                             void* gm, int64_t sgm, void* ga, int64_t sga, int64_t B, int64_t F, int pre_tanh,
                             hipStream_t st) {
  constexpr int VN = VecOf<T>::n;
  const bool vec = (F % VN == 0) && (sa % VN == 0) && (sgm % VN == 0) && (sga % VN == 0) && aligned16(u) &&
                   aligned16(a) && (!gu || aligned16(gu)) && (!gx || aligned16(gx)) && aligned16(gm) && aligned16(ga);
  const RowGeom g = row_geom(B, vec ? F / VN : F, kGridStream);
  if (vec) {
    VAE_SWITCH_U(g.U, hipLaunchKernelGGL((made_tail_bwd_kernel<T, VN, U>), dim3(g.grid), dim3(kBlock), 0, st,
                                         (const T*)u, (const T*)a, sa, (const T*)gu, (const T*)gld, (T*)gx, (T*)gm, sgm,
                                         (T*)ga, sga, B, (int)F, pre_tanh, g.G, g.tiles));
  } else {
    VAE_SWITCH_U(g.U, hipLaunchKernelGGL((made_tail_bwd_kernel<T, 1, U>), dim3(g.grid), dim3(kBlock), 0, st,
                                         (const T*)u, (const T*)a, sa, (const T*)gu, (const T*)gld, (T*)gx, (T*)gm, sgm,
                                         (T*)ga, sga, B, (int)F, pre_tanh, g.G, g.tiles));
  }
  return hipGetLastError() == hipSuccess ? VAE_OK : VAE_ERR_LAUNCH;
}

template <typename T>
static int coupling_fwd_dispatch(const void* x, const void* s_raw, const void* t_raw, const float* mask, void* y,
                                 const void* ld_in, void* ld_out, int64_t B, int64_t F, hipStream_t st) {
  constexpr int VN = VecOf<T>::n;
  const bool vec = (F % VN == 0) && aligned16(x) && (!s_raw || aligned16(s_raw)) && aligned16(t_raw) &&
                   aligned16(mask) && aligned16(y);
  const RowGeom g = row_geom(B, vec ? F / VN : F, kGridStream);
  if (vec) {
    VAE_SWITCH_U(g.U, hipLaunchKernelGGL((coupling_tail_fwd_kernel<T, VN, U>), dim3(g.grid), dim3(kBlock), 0, st,
                                         (const T*)x, (const T*)s_raw, (const T*)t_raw, mask, (T*)y, (const T*)ld_in,
                                         (T*)ld_out, B, (int)F, g.G, g.tiles));
  } else {
    VAE_SWITCH_U(g.U, hipLaunchKernelGGL((coupling_tail_fwd_kernel<T, 1, U>), dim3(g.grid), dim3(kBlock), 0, st,
                                         (const T*)x, (const T*)s_raw, (const T*)t_raw, mask, (T*)y, (const T*)ld_in,
                                         (T*)ld_out, B, (int)F, g.G, g.tiles));
  }
  return hipGetLastError() == hipSuccess ? VAE_OK : VAE_ERR_LAUNCH;
}

template <typename T>
static int coupling_bwd_dispatch(const void* x, const void* s_raw, const float* mask, const void* gy, const void* gld,
                                 void* gx, void* gs, void* gt, int64_t B, int64_t F, hipStream_t st) {
  constexpr int VN = VecOf<T>::n;
  const bool vec = (F % VN == 0) && aligned16(x) && (!s_raw || aligned16(s_raw)) && aligned16(mask) &&
                   (!gy || aligned16(gy)) && (!gx || aligned16(gx)) && (!gs || aligned16(gs)) && aligned16(gt);
  const RowGeom g = row_geom(B, vec ? F / VN : F, kGridStream);
  if (vec) {
    VAE_SWITCH_U(g.U, hipLaunchKernelGGL((coupling_tail_bwd_kernel<T, VN, U>), dim3(g.grid), dim3(kBlock), 0, st,
                                         (const T*)x, (const T*)s_raw, mask, (const T*)gy, (const T*)gld, (T*)gx,
                                         (T*)gs, (T*)gt, B, (int)F, g.G, g.tiles));
  } else {
    VAE_SWITCH_U(g.U, hipLaunchKernelGGL((coupling_tail_bwd_kernel<T, 1, U>), dim3(g.grid), dim3(kBlock), 0, st,
                                         (const T*)x, (const T*)s_raw, mask, (const T*)gy, (const T*)gld, (T*)gx,
                                         (T*)gs, (T*)gt, B, (int)F, g.G, g.tiles));
  }
  return hipGetLastError() == hipSuccess ? VAE_OK : VAE_ERR_LAUNCH;
}

// stats layout (floats, caller-owned, >= vae_bnflow_stats_floats(F)):
//   [0,F) mean | [F,2F) var | [2F,3F) rstd | [3F,4F) scale | [4F,5F) beta | [5F,6F) ldc |
//   [6F,7F) S1 | [7F,8F) S2 | [8F] sum gld | pad to 8F+4 | [8F+4, 9F+4) k1 | [9F+4, 10F+4) K |
//   block partials kBnGrid * (2F+1)
static size_t bn_stats_floats(int64_t F) { return 10 * (size_t)F + 4 + (size_t)kBnGrid * (2 * (size_t)F + 1); }
constexpr int kBnApplyGrid = 2048;

template <typename T>
static int bn_fwd_dispatch(const void* x, const float* lg, const float* beta, const float* mean_in, const float* var_in,
                           float* stats, void* y, const void* ld_in, void* ld_out, int64_t B, int64_t F, float eps,
                           int training, int rev, hipStream_t st) {
  constexpr int VN = VecOf<T>::n;
  const bool vec = (F % VN == 0) && aligned16(x) && (!y || aligned16(y)) && aligned16(stats);
  float *mean = stats, *scale = stats + 3 * F, *shift = stats + 4 * F, *ldc = stats + 5 * F,
        *part = stats + 10 * F + 4;
  if (training) {
    int G, NJ, grid;
    int64_t tiles;
    bn_geom(B, vec ? F / VN : F, &G, &NJ, &tiles, &grid);
    if (training == 2) {
      // the block partials are already in `stats`: vae_made_tail_fwd_moments wrote them (same B, F -> same grid)
      if (!vec) return VAE_ERR_ARG;
    } else if (vec)
      bn_colstats_launch<T, VN>(NJ, grid, st, (const T*)x, nullptr, nullptr, nullptr, nullptr, part, B, (int)F, 0, 0, G,
                                tiles);
    else
      bn_colstats_launch<T, 1>(NJ, grid, st, (const T*)x, nullptr, nullptr, nullptr, nullptr, part, B, (int)F, 0, 0, G,
                               tiles);
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((unsigned)((F + kFinCols - 1) / kFinCols)), dim3(kBlock), 0, st, part,
                       grid, x, DTypeCode<T>::value, lg, beta, stats, B, (int)F, eps, 0, 1);
    if (!y) return hipGetLastError() == hipSuccess ? VAE_OK : VAE_ERR_LAUNCH;  // statistics only
  } else {
    hipLaunchKernelGGL(bn_prepare_kernel, dim3((unsigned)((F + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, mean_in,
                       var_in, lg, beta, stats, (int)F);
  }
  const RowGeom g = row_geom(B, vec ? F / VN : F, kBnApplyGrid);
  const bool hoist = (vec ? F / VN : F) <= (int64_t)g.G * g.U;
#define VAE_BN_FWD(VV, HH)                                                                                            \
  VAE_SWITCH_U(g.U, hipLaunchKernelGGL((bn_apply_fwd_kernel<T, VV, U, HH>), dim3(g.grid), dim3(kBlock), 0, st,        \
                                       (const T*)x, mean, scale, shift, ldc, (T*)y, (const T*)ld_in, (T*)ld_out, B,   \
                                       (int)F, rev, g.G, g.tiles))
  if (vec) {
    if (hoist) { VAE_BN_FWD(VN, true); } else { VAE_BN_FWD(VN, false); }
  } else {
    if (hoist) { VAE_BN_FWD(1, true); } else { VAE_BN_FWD(1, false); }
  }
#undef VAE_BN_FWD
  return hipGetLastError() == hipSuccess ? VAE_OK : VAE_ERR_LAUNCH;
}

template <typename T>
static int bn_bwd_dispatch(const void* x, const void* gy, const void* gld, float* stats, void* gx, int64_t B, int64_t F,
                           int training, int rev, int phase, int64_t count, hipStream_t st) {
  constexpr int VN = VecOf<T>::n;
  const bool vec = (F % VN == 0) && aligned16(x) && aligned16(gy) && (!gx || aligned16(gx)) && aligned16(stats);
  float *mean = stats, *rstd = stats + 2 * F, *scale = stats + 3 * F, *k1 = stats + 8 * F + 4, *kk = stats + 9 * F + 4,
        *part = stats + 10 * F + 4;
  if (phase != 2) {
    int G, NJ, grid;
    int64_t tiles;
    bn_geom(B, vec ? F / VN : F, &G, &NJ, &tiles, &grid);
    if (vec)
      bn_colstats_launch<T, VN>(NJ, grid, st, (const T*)x, (const T*)gy, (const T*)gld, mean, rstd, part, B, (int)F, 1,
                                rev, G, tiles);
    else
      bn_colstats_launch<T, 1>(NJ, grid, st, (const T*)x, (const T*)gy, (const T*)gld, mean, rstd, part, B, (int)F, 1,
                               rev, G, tiles);
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((unsigned)((F + kFinCols - 1) / kFinCols)), dim3(kBlock), 0, st, part,
                       grid, (const void*)nullptr, 0, (const float*)nullptr, (const float*)nullptr, stats, B, (int)F, 0.f,
                       1, training);
    if (phase == 1) return hipGetLastError() == hipSuccess ? VAE_OK : VAE_ERR_LAUNCH;  // the three sums only
  } else {
    hipLaunchKernelGGL(bn_bwd_consts_kernel, dim3((unsigned)((F + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, stats,
                       (int)F, count);
  }
  const RowGeom g = row_geom(B, vec ? F / VN : F, kBnApplyGrid);
  const bool hoist = (vec ? F / VN : F) <= (int64_t)g.G * g.U;
#define VAE_BN_BWD(VV, HH)                                                                                            \
  VAE_SWITCH_U(g.U, hipLaunchKernelGGL((bn_apply_bwd_kernel<T, VV, U, HH>), dim3(g.grid), dim3(kBlock), 0, st,        \
                                       (const T*)x, (const T*)gy, mean, scale, k1, kk, (T*)gx, B, (int)F, training,   \
                                       rev, g.G, g.tiles))
  if (vec) {
    if (hoist) { VAE_BN_BWD(VN, true); } else { VAE_BN_BWD(VN, false); }
  } else {
    if (hoist) { VAE_BN_BWD(1, true); } else { VAE_BN_BWD(1, false); }
  }
#undef VAE_BN_BWD
  return hipGetLastError() == hipSuccess ? VAE_OK : VAE_ERR_LAUNCH;
}

}  // namespace vae

using namespace vae;

extern "C" {

int vae_made_tail_fwd(const void* x, int64_t stride_x, const void* m, int64_t stride_m, const void* a, int64_t stride_a,
                      void* u, const void* logdet_in, void* logdet_out, int64_t B, int64_t F, int pre_exp_tanh,
                      int dtype, vae_stream_t stream) {
  if (B < 0 || F <= 0 || F > VAE_MAX_ROW) return VAE_ERR_ARG;
  if (B == 0) return VAE_OK;
  if (!x || !m || !a || !u || !logdet_out) return VAE_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == VAE_F32)
    return made_fwd_dispatch<float>(x, stride_x, m, stride_m, a, stride_a, u, logdet_in, logdet_out, B, F, pre_exp_tanh, st);
  if (dtype == VAE_BF16)
    return made_fwd_dispatch<bf16_t>(x, stride_x, m, stride_m, a, stride_a, u, logdet_in, logdet_out, B, F, pre_exp_tanh, st);
  if (dtype == VAE_F16)
    return made_fwd_dispatch<f16_t>(x, stride_x, m, stride_m, a, stride_a, u, logdet_in, logdet_out, B, F, pre_exp_tanh, st);
  return VAE_ERR_ARG;
}

int vae_made_tail_moments_supported(int64_t F, int dtype) {
  if (dtype != VAE_F32 && dtype != VAE_BF16 && dtype != VAE_F16) return 0;
  const int VN = dtype == VAE_F32 ? 4 : 8;
  if (F <= 0 || F % VN) return 0;
  int G, NJ, grid;
  int64_t tiles;
  bn_geom(1, F / VN, &G, &NJ, &tiles, &grid);
  return (int64_t)G * kBnChunks >= F / VN ? 1 : 0;
}

int vae_made_tail_fwd_moments(const void* x, int64_t stride_x, const void* m, int64_t stride_m, const void* a,
                              int64_t stride_a, void* u, const void* logdet_in, void* logdet_out, float* bn_stats,
                              int64_t B, int64_t F, int pre_exp_tanh, int dtype, vae_stream_t stream) {
  if (B < 0 || F <= 0 || F > VAE_MAX_ROW) return VAE_ERR_ARG;
  if (B == 0) return VAE_OK;
  if (!x || !m || !a || !u || !logdet_out || !bn_stats) return VAE_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == VAE_F32)
    return made_moments_dispatch<float>(x, stride_x, m, stride_m, a, stride_a, u, logdet_in, logdet_out, bn_stats, B, F,
                                        pre_exp_tanh, st);
  if (dtype == VAE_BF16)
    return made_moments_dispatch<bf16_t>(x, stride_x, m, stride_m, a, stride_a, u, logdet_in, logdet_out, bn_stats, B, F,
                                         pre_exp_tanh, st);
  if (dtype == VAE_F16)
    return made_moments_dispatch<f16_t>(x, stride_x, m, stride_m, a, stride_a, u, logdet_in, logdet_out, bn_stats, B, F,
                                        pre_exp_tanh, st);
  return VAE_ERR_ARG;
}

int vae_made_tail_bwd(const void* u, const void* a, int64_t stride_a, const void* gu, const void* glogdet, void* gx,
                      void* gm, int64_t stride_gm, void* ga, int64_t stride_ga, int64_t B, int64_t F, int pre_exp_tanh,
                      int dtype, vae_stream_t stream) {
  if (B < 0 || F <= 0 || F > VAE_MAX_ROW) return VAE_ERR_ARG;
  if (B == 0) return VAE_OK;
  if (!u || !a || !gm || !ga) return VAE_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == VAE_F32)
    return made_bwd_dispatch<float>(u, a, stride_a, gu, glogdet, gx, gm, stride_gm, ga, stride_ga, B, F, pre_exp_tanh, st);
  if (dtype == VAE_BF16)
    return made_bwd_dispatch<bf16_t>(u, a, stride_a, gu, glogdet, gx, gm, stride_gm, ga, stride_ga, B, F, pre_exp_tanh, st);
  if (dtype == VAE_F16)
    return made_bwd_dispatch<f16_t>(u, a, stride_a, gu, glogdet, gx, gm, stride_gm, ga, stride_ga, B, F, pre_exp_tanh, st);
  return VAE_ERR_ARG;
}

int vae_coupling_tail_fwd(const void* x, const void* s_raw, const void* t_raw, const float* mask, void* y,
                          const void* logdet_in, void* logdet_out, int64_t B, int64_t F, int dtype,
                          vae_stream_t stream) {
  if (B < 0 || F <= 0 || F > VAE_MAX_ROW) return VAE_ERR_ARG;
  if (B == 0) return VAE_OK;
  if (!x || !t_raw || !mask || !y || !logdet_out) return VAE_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == VAE_F32) return coupling_fwd_dispatch<float>(x, s_raw, t_raw, mask, y, logdet_in, logdet_out, B, F, st);
  if (dtype == VAE_BF16) return coupling_fwd_dispatch<bf16_t>(x, s_raw, t_raw, mask, y, logdet_in, logdet_out, B, F, st);
  if (dtype == VAE_F16) return coupling_fwd_dispatch<f16_t>(x, s_raw, t_raw, mask, y, logdet_in, logdet_out, B, F, st);
  return VAE_ERR_ARG;
}

int vae_coupling_tail_bwd(const void* x, const void* s_raw, const float* mask, const void* gy, const void* glogdet,
                          void* gx, void* gs_raw, void* gt_raw, int64_t B, int64_t F, int dtype, vae_stream_t stream) {
  if (B < 0 || F <= 0 || F > VAE_MAX_ROW) return VAE_ERR_ARG;
  if (B == 0) return VAE_OK;
  if (!x || !mask || !gt_raw || (s_raw && !gs_raw)) return VAE_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == VAE_F32) return coupling_bwd_dispatch<float>(x, s_raw, mask, gy, glogdet, gx, gs_raw, gt_raw, B, F, st);
  if (dtype == VAE_BF16) return coupling_bwd_dispatch<bf16_t>(x, s_raw, mask, gy, glogdet, gx, gs_raw, gt_raw, B, F, st);
  if (dtype == VAE_F16) return coupling_bwd_dispatch<f16_t>(x, s_raw, mask, gy, glogdet, gx, gs_raw, gt_raw, B, F, st);
  return VAE_ERR_ARG;
}

size_t vae_bnflow_stats_floats(int64_t F) { return F > 0 ? bn_stats_floats(F) : 0; }

int vae_bnflow_fwd(const void* x, const float* log_gamma, const float* beta, const float* mean_in, const float* var_in,
                   float* stats, void* y, const void* logdet_in, void* logdet_out, int64_t B, int64_t F, float eps,
                   int training, int reverse, int dtype, vae_stream_t stream) {
  if (B < 0 || F <= 0 || F > VAE_MAX_ROW) return VAE_ERR_ARG;
  if (B == 0) return VAE_OK;
  if (!x || !log_gamma || !beta || !stats) return VAE_ERR_ARG;
  if ((!y || !logdet_out) && !(training && !y)) return VAE_ERR_ARG;  // y == NULL: training statistics only
  if (!training && (!mean_in || !var_in)) return VAE_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == VAE_F32)
    return bn_fwd_dispatch<float>(x, log_gamma, beta, mean_in, var_in, stats, y, logdet_in, logdet_out, B, F, eps,
                                  training, reverse, st);
  if (dtype == VAE_BF16)
    return bn_fwd_dispatch<bf16_t>(x, log_gamma, beta, mean_in, var_in, stats, y, logdet_in, logdet_out, B, F, eps,
                                   training, reverse, st);
  if (dtype == VAE_F16)
    return bn_fwd_dispatch<f16_t>(x, log_gamma, beta, mean_in, var_in, stats, y, logdet_in, logdet_out, B, F, eps,
                                   training, reverse, st);
  return VAE_ERR_ARG;
}

int vae_bnflow_bwd(const void* x, const void* gy, const void* glogdet, float* stats, void* gx, int64_t B, int64_t F,
                   int training, int reverse, int dtype, vae_stream_t stream) {
  if (B < 0 || F <= 0 || F > VAE_MAX_ROW) return VAE_ERR_ARG;
  if (B == 0) return VAE_OK;
  if (!x || !gy || !stats || !gx) return VAE_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == VAE_F32) return bn_bwd_dispatch<float>(x, gy, glogdet, stats, gx, B, F, training, reverse, 0, B, st);
  if (dtype == VAE_BF16) return bn_bwd_dispatch<bf16_t>(x, gy, glogdet, stats, gx, B, F, training, reverse, 0, B, st);
  if (dtype == VAE_F16) return bn_bwd_dispatch<f16_t>(x, gy, glogdet, stats, gx, B, F, training, reverse, 0, B, st);
  return VAE_ERR_ARG;
}

int vae_bnflow_bwd_sums(const void* x, const void* gy, const void* glogdet, float* stats, int64_t B, int64_t F,
                        int reverse, int dtype, vae_stream_t stream) {
  if (B < 0 || F <= 0 || F > VAE_MAX_ROW) return VAE_ERR_ARG;
  if (B == 0) return VAE_OK;
  if (!x || !gy || !stats) return VAE_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == VAE_F32) return bn_bwd_dispatch<float>(x, gy, glogdet, stats, nullptr, B, F, 1, reverse, 1, B, st);
  if (dtype == VAE_BF16) return bn_bwd_dispatch<bf16_t>(x, gy, glogdet, stats, nullptr, B, F, 1, reverse, 1, B, st);
  if (dtype == VAE_F16) return bn_bwd_dispatch<f16_t>(x, gy, glogdet, stats, nullptr, B, F, 1, reverse, 1, B, st);
  return VAE_ERR_ARG;
}

int vae_bnflow_bwd_apply(const void* x, const void* gy, float* stats, void* gx, int64_t B, int64_t F, int64_t count,
                         int reverse, int dtype, vae_stream_t stream) {
  if (B < 0 || F <= 0 || F > VAE_MAX_ROW || count < B || count <= 0) return VAE_ERR_ARG;
  if (B == 0) return VAE_OK;
  if (!x || !gy || !stats || !gx) return VAE_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == VAE_F32) return bn_bwd_dispatch<float>(x, gy, nullptr, stats, gx, B, F, 1, reverse, 2, count, st);
  if (dtype == VAE_BF16) return bn_bwd_dispatch<bf16_t>(x, gy, nullptr, stats, gx, B, F, 1, reverse, 2, count, st);
  if (dtype == VAE_F16) return bn_bwd_dispatch<f16_t>(x, gy, nullptr, stats, gx, B, F, 1, reverse, 2, count, st);
  return VAE_ERR_ARG;
}

}  // extern "C"
