// The three per-element NLLs (value and d/d(recon)), shared by elbo.hip and multi.hip.
// Definitions: this build's stated contract for the un-vendored helpers.distributions.nll
// ("parity unpinned", see oracle/cpu_ref.py).
#pragma once
#include "common.hpp"
#include "../../include/vae_amd.h"

namespace vae {

constexpr float kHalfLog2Pi = 0.91893853320467274178f;
constexpr float kBin = 1.0f / 256.0f;

__device__ __forceinline__ float sigmoidf_(float v) { return fast_rcp(1.0f + __expf(-v)); }
// sigmoid for the discretized logistic, whose two CDF values are subtracted from each other (an
// ulp there is amplified by 1/P): accurate exp, near-correctly-rounded reciprocal.
__device__ __forceinline__ float sigmoid_acc(float v) { return rcp_nr(1.0f + exp_acc(-v)); }
// softplus(v) = log(1 + e^v): linear above 20 (torch.nn.functional.softplus), series for tiny e^v
__device__ __forceinline__ float softplusf_(float v) {
  const float e = __expf(fminf(v, 20.0f));
  return v > 20.0f ? v : log1p_unit_any(e);
}

// ---- the three NLLs: value and d/d(recon) per element --------------------------
// NLL == VAE_NLL_BERNOULLI: -Bernoulli(logits=l).log_prob(x) = max(l,0) - l*x + log1p(exp(-|l|))
// The softplus term uses v_exp_f32 / v_log_f32 directly: its absolute error (<= ~1e-7 per element)
// is below the rounding of max(l,0) - l*x itself, and the exact ocml forms cost 15 % of this
// kernel's time (measured: 4.5 -> 5.1 TB/s).
__device__ __forceinline__ float nll_bern(float x, float l) {
  return fmaxf(l, 0.0f) - l * x + fast_log(1.0f + fast_exp(-fabsf(l)));
}
__device__ __forceinline__ float nll_bern_grad(float x, float l) { return fast_rcp(1.0f + __expf(-l)) - x; }

// Value AND gradient of V elements (V even), for the one-pass kernels.  At 2 bytes per element those are no longer
// HBM-bound but co-bound by VALU issue (r03: 406 instructions per wave, ~25 per element), so the arithmetic is laid
// out for the packed-fp32 pipe (v_pk_add / v_pk_mul / v_pk_fma_f32: two elements per lane and instruction):
//   e = exp(-|l|), p = 1 + e, r = 1/p, pos = [sign bit of l clear], d = pos - x
//   value    = max(l,0) - l x + log1p(e) = l d + ln2 log2(p)                         (two packed fma into the sum)
//   gradient = sigmoid(l) - x            = d - copysign(e r, l)    (l >= 0: (1 - x) - e r; l < 0: e r - x: no
//                                                                   cancellation at x = 1, as before)
// One exponential serves both; `pos` and the copysign both read the SIGN BIT of l, so l = -0.0 is consistent
// (pos = 0, -e r = -1/2: gradient 1/2 - x).  Per element: 3 quarter-rate transcendentals + ~10 full-rate
// instruction slots (was ~22).  `s2` carries two partial sums (even / odd elements), added by the caller.
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int V>
__device__ __forceinline__ void nll_bern_value_grad(const float (&x)[V], const float (&l)[V], float cscale, f32x2& s2,
                                                    float (&g)[V]) {
  static_assert(V == 1 || V % 2 == 0, "pairs, or the scalar fallback kernels' single element");
  if constexpr (V == 1) {
    const float e = __builtin_amdgcn_exp2f(-fabsf(l[0]) * kLog2e), p = 1.0f + e;
    const float d = (__builtin_bit_cast(int, l[0]) < 0 ? 0.0f : 1.0f) - x[0];
    s2[0] = __builtin_fmaf(kLn2, __builtin_amdgcn_logf(p), __builtin_fmaf(l[0], d, s2[0]));
    g[0] = (d - __builtin_copysignf(e * __builtin_amdgcn_rcpf(p), l[0])) * cscale;
    return;
  }
  const f32x2 one = {1.0f, 1.0f}, ln2 = {kLn2, kLn2}, cs = {cscale, cscale};
#pragma unroll
  for (int i = 0; i + 1 < V; i += 2) {
    const f32x2 lv = {l[i], l[i + 1]}, xv = {x[i], x[i + 1]};
    // (element-wise on the scalars, not through lv[k]: hipcc 7.2 folded `bit_cast<int>(lv[k]) < 0` of both halves
    // into the compare of the LOW half -- op_sel_hi:[0,1] on the packed subtract -- and every odd element took its
    // neighbour's sign)
    const float p0 = __builtin_bit_cast(int, l[i]) < 0 ? 0.0f : 1.0f;
    const float p1 = __builtin_bit_cast(int, l[i + 1]) < 0 ? 0.0f : 1.0f;
    const f32x2 pos = {p0, p1};
    const f32x2 e = {__builtin_amdgcn_exp2f(-fabsf(l[i]) * kLog2e), __builtin_amdgcn_exp2f(-fabsf(l[i + 1]) * kLog2e)};
    const f32x2 p = one + e;
    f32x2 r, lg;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      r[k] = __builtin_amdgcn_rcpf(p[k]);
      lg[k] = __builtin_amdgcn_logf(p[k]);
    }
    const f32x2 er = e * r;
    const f32x2 d = pos - xv;
    s2 = __builtin_elementwise_fma(lv, d, s2);
    s2 = __builtin_elementwise_fma(ln2, lg, s2);
    f32x2 ser;
#pragma unroll
    for (int k = 0; k < 2; ++k) ser[k] = __builtin_copysignf(er[k], lv[k]);
    const f32x2 gg = (d - ser) * cs;
    g[i] = gg[0];
    g[i + 1] = gg[1];
  }
}

// NLL == VAE_NLL_GAUSSIAN: -Normal(m, softplus(v) + 1e-6).log_prob(x)
__device__ __forceinline__ float nll_gauss(float x, float m, float v) {
  const float s = softplusf_(v) + 1e-6f;
  const float d = (x - m) * rcp_nr(s);
  return 0.5f * d * d + fast_log(s) + kHalfLog2Pi;
}
__device__ __forceinline__ void nll_gauss_grad(float x, float m, float v, float& gm, float& gv) {
  const float s = softplusf_(v) + 1e-6f;
  const float is = rcp_nr(s);
  const float d = (x - m) * is;
  gm = -d * is;
  gv = (is - d * d * is) * sigmoidf_(v);
}

// (Tried, r01: forming sigmoid(tp) - sigmoid(t) as E (1 - d) / ((1 + E d)(1 + E)) without the subtraction is
// 10 points faster in the forward and far better conditioned, but it no longer reproduces the fp32 reference where
// both sigmoids round to within a few ulp of 1 -- there the reference's value IS its rounding pattern (multiples of
// 6e-8 against the 1e-7 floor) and row NLLs moved by 1e-4..2e-3 relative.  Parity wins: two rounded sigmoids.)
// NLL == VAE_NLL_DISC_LOGISTIC (256 bins): xq = floor(x*256)/256, m = sigmoid(a),
// lv = clamp(v, -4.5, 0), 1/sc = exp(-lv), t = (xq - m)/sc,
// -log(sigmoid(t + bin/sc) - sigmoid(t) + 1e-7)
__device__ __forceinline__ float nll_dlog(float x, float a, float v) {
  const float xq = floorf(x * 256.0f) * kBin;
  const float m = sigmoidf_(a);
  const float isc = __expf(-fminf(fmaxf(v, -4.5f), 0.0f));
  const float t = (xq - m) * isc;
  const float cp = sigmoid_acc(__builtin_fmaf(kBin, isc, t));
  const float cm = sigmoid_acc(t);
  return -fast_log(cp - cm + 1e-7f);
}
__device__ __forceinline__ void nll_dlog_grad(float x, float a, float v, float& ga, float& gv) {
  const float xq = floorf(x * 256.0f) * kBin;
  const float m = sigmoidf_(a);
  const float isc = __expf(-fminf(fmaxf(v, -4.5f), 0.0f));
  const float t = (xq - m) * isc;
  const float tp = __builtin_fmaf(kBin, isc, t);
  const float cp = sigmoid_acc(tp);
  const float cm = sigmoid_acc(t);
  const float ip = rcp_nr(cp - cm + 1e-7f);
  const float dp = -cp * (1.0f - cp) * ip;  // d nll / d tp
  const float dm = cm * (1.0f - cm) * ip;   // d nll / d t
  ga = (dp + dm) * (-isc) * m * (1.0f - m);
  const float inside = (v > -4.5f && v < 0.0f) ? 1.0f : 0.0f;  // hardtanh passes no gradient at the clamps
  gv = (dp * (-tp) + dm * (-t)) * inside;
}

// Where the rows of an NLL launch live.  FlatRows: one [B, D] target and one [B, D | 2D] decoder output (and its
// gradient).  StepRows: T such pairs given by pointer -- the per-step reconstructions and targets of a VRNN
// (vrnn.py:805-829) as the decoder left them, no torch.cat copy -- indexed as one [T*B] batch, step-major; a tile
// never straddles two steps, so the step (and with it the table lookup) is uniform over a block.
template <typename T>
struct FlatRows {
  const T* x;
  const T* recon;
  T* grecon;
  int64_t B;
  __device__ __forceinline__ bool locate(int64_t tile, int rpb, int rloc, int64_t D, int64_t rstride, int64_t& row,
                                         const T*& xr, const T*& rr, T*& gr) const {
    row = tile * rpb + rloc;
    if (row >= B) return false;
    xr = x + row * D;
    rr = recon + row * rstride;
    gr = grecon + row * rstride;
    return true;
  }
  // the KL weight of the tile's rows: one value for the whole launch
  __device__ __forceinline__ float beta(int64_t, float kl_beta) const { return kl_beta; }
};
// `beta`: the KL weight of each step.  The reference calls compute_kl_beta once per time step
// (vrnn.py:825-827 -> abstract_vae.py:363), so a stateful annealer hands every step its own value.
struct StepTable {
  const void* x[VAE_MAX_STEPS];
  const void* recon[VAE_MAX_STEPS];
  float beta[VAE_MAX_STEPS];
};
struct StepTableG {
  const void* x[VAE_MAX_STEPS];
  const void* recon[VAE_MAX_STEPS];
  void* grecon[VAE_MAX_STEPS];
  float beta[VAE_MAX_STEPS];
};
template <typename T, typename Tab>
struct StepRows {
  Tab tab;
  int64_t B;     // rows per step
  int64_t tps;   // tiles per step = ceil(B / rows_per_block)
  __device__ __forceinline__ bool locate(int64_t tile, int rpb, int rloc, int64_t D, int64_t rstride, int64_t& row,
                                         const T*& xr, const T*& rr, T*& gr) const {
    const int64_t step = tile / tps;
    const int64_t b = (tile - step * tps) * rpb + rloc;
    row = step * B + b;
    if (b >= B) return false;
    xr = static_cast<const T*>(tab.x[step]) + b * D;
    rr = static_cast<const T*>(tab.recon[step]) + b * rstride;
    if constexpr (sizeof(Tab) == sizeof(StepTableG)) gr = static_cast<T*>(tab.grecon[step]) + b * rstride;
    else gr = nullptr;
    return true;
  }
  __device__ __forceinline__ float beta(int64_t tile, float) const { return tab.beta[tile / tps]; }
};

// Long rows with few samples: give a row to the whole block so the chip stays full.
// `long_rows_always`: the store-heavy backward of 3x64x64 rows measured 3-4 points faster block-per-row at any B
// (fewer concurrent DRAM streams: 5 per block instead of 5 per wave); the forward is not (r01, B=8192).
// `loads_only` (the forward kernels: nothing but [B] vectors is stored): rows of 65..128 vectors run two rows per wave
// (G = 32, four chunks per lane) instead of one row per wave with a half-empty second chunk -- bf16 784-pixel rows
// (98 vectors) 57.6 -> 67 % of peak (r02, scripts/ab.py); the store-heavy backward keeps the wave per row.
inline RowGeom nll_geom(int64_t B, int64_t vecs_per_row, int64_t max_grid, bool long_rows_always = false,
                        bool loads_only = false) {
  RowGeom g = row_geom(B, vecs_per_row, max_grid);
  if (loads_only && g.G == 64 && vecs_per_row <= 128) regroup(g, B, 32, 4, max_grid);
  if (vecs_per_row >= 1024 && (long_rows_always || g.tiles < 4 * 256) && !knobs().G) {
    g.G = kBlock;
    g.U = knobs().U ? knobs().U : 4;
    g.rows_per_block = 1;
    g.tiles = B;
    g.grid = (int)(g.tiles < max_grid ? g.tiles : max_grid);
  }
  return g;
}

// The six batch means (abstract_vae.py:381-387) from the per-sample vectors the NLL + ELBO forward has just written
// (loss, elbo, nll) and read (kld, add, mi): a closing launch of <= 64 blocks behind the streaming kernel.  The
// streaming kernels then carry no reduction at all -- no accumulators, no barrier, no ticket -- and run one row tile
// per block, handed out by the hardware dispatcher as blocks retire.  Measured r02 in the bench.py step (B = 65536,
// 784 px, one-pass NLL, event-bracketed): streaming kernel alone 104.3 us; + this launch 111.5 us; against 115.3 us
// for <= 1024 persistent blocks with grid_reduce inside the kernel, 119-121 us for one tile per block with a
// per-block or two-level ticket inside the kernel (the closing chain -- store drain, ticket(s), acquire, partial
// loads -- is 8-10 dependent round trips of ~1 us on a loaded chip; a static partition loses ~7 us to the dynamic
// hand-out whatever its grid: scripts/membench.hip).  Thread i adds vectors i, i + grid*256, ... in order (double);
// blocks in block order: deterministic for a given B.
template <typename T, int V, int kAhead = 4>
__global__ __launch_bounds__(kBlock) void elbo_rows_means_kernel(const T* __restrict__ loss, const T* __restrict__ elbo,
                                                                 const T* __restrict__ nll, const T* __restrict__ kld,
                                                                 const T* __restrict__ add, const T* __restrict__ mi,
                                                                 float* __restrict__ ws, float inv_count,
                                                                 float* __restrict__ means, int64_t B, int64_t steps,
                                                                 T* __restrict__ loss_avg, T* __restrict__ elbo_avg,
                                                                 T* __restrict__ elbo_lse) {
  // per-sample averages over the steps of a [steps * Bs] launch (vrnn.py:767-803: the per-step [B] vectors are
  // summed one after the other in the tensor dtype, then divided by the step count)
  if (loss_avg) {
    const int64_t Bs = B / steps;
    const float inv_steps = div_rn(1.0f, (float)steps);
    for (int64_t b = (int64_t)blockIdx.x * kBlock + threadIdx.x; b < Bs; b += (int64_t)gridDim.x * kBlock) {
      float sl = to_f32(loss[b]), se = to_f32(elbo[b]);
      // eight steps' values requested together (the additions stay in step order: their roundings are the reference's)
      for (int64_t t0 = 1; t0 < steps; t0 += 8) {
        float vl[8], ve[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int64_t t = t0 + k < steps ? t0 + k : steps - 1;
          vl[k] = to_f32(loss[t * Bs + b]);
          ve[k] = to_f32(elbo[t * Bs + b]);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          if (t0 + k < steps) {
            sl = to_f32(from_f32<T>(add_rn(sl, vl[k])));
            se = to_f32(from_f32<T>(add_rn(se, ve[k])));
          }
        }
      }
      // tensor /= python_int on the device is a multiplication by the fp32 reciprocal in stock PyTorch (div_true
      // with a CPU scalar divisor): the same here
      loss_avg[b] = from_f32<T>(mul_rn(sl, inv_steps));
      elbo_avg[b] = from_f32<T>(mul_rn(se, inv_steps));
    }
  }
  // per-sample log-sum-exp over the steps of the elbo column: the importance-sampling estimate of
  // abstract_vae.py:311-314 when the "steps" are draws of the same posterior (max first, as torch.logsumexp does)
  if (elbo_lse) {
    const int64_t Bs = B / steps;
    for (int64_t b = (int64_t)blockIdx.x * kBlock + threadIdx.x; b < Bs; b += (int64_t)gridDim.x * kBlock) {
      float m = to_f32(elbo[b]);
#pragma unroll 8
      for (int64_t t = 1; t < steps; ++t) m = fmaxf(m, to_f32(elbo[t * Bs + b]));
      const float shift = (fabsf(m) == INFINITY) ? 0.0f : m;
      float s = 0.0f;
#pragma unroll 8
      for (int64_t t = 0; t < steps; ++t) s += expf(to_f32(elbo[t * Bs + b]) - shift);
      elbo_lse[b] = from_f32<T>(logf(s) + shift);
    }
  }
  double v[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  const int64_t nvec = B / V;   // whole vectors; the B % V rows of the tail go to the first threads of block 0
  // kAhead of a thread's vectors are requested before the first is added (the single-block form of small batches walks
  // up to 8 per array: one memory round trip per vector otherwise); the additions stay in j order
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t j0 = (int64_t)blockIdx.x * kBlock + threadIdx.x; j0 < nvec; j0 += stride * kAhead) {
    float a[kAhead][6][V];
#pragma unroll
    for (int u = 0; u < kAhead; ++u) {
      const int64_t j = j0 + u * stride < nvec ? j0 + u * stride : j0;
      load_vec<T, V, false>(loss + j * V, a[u][0]);
      load_vec<T, V, false>(elbo + j * V, a[u][1]);
      load_vec<T, V, false>(nll + j * V, a[u][2]);
      if (kld) load_vec<T, V, false>(kld + j * V, a[u][3]);
      if (add) load_vec<T, V, false>(add + j * V, a[u][4]);
      if (mi) load_vec<T, V, false>(mi + j * V, a[u][5]);
    }
#pragma unroll
    for (int u = 0; u < kAhead; ++u) {
      if (j0 + u * stride < nvec) {
#pragma unroll
        for (int i = 0; i < V; ++i) {
          v[0] += (double)a[u][0][i];
          v[1] += (double)a[u][1][i];
          v[2] += (double)a[u][2][i];
          if (kld) v[3] += (double)a[u][3][i];
          if (add) v[4] += (double)a[u][4][i];
          if (mi) v[5] += (double)a[u][5][i];
        }
      }
    }
  }
  if (blockIdx.x == 0 && (int64_t)threadIdx.x < B - nvec * V) {
    const int64_t row = nvec * V + threadIdx.x;
    v[0] += (double)to_f32(loss[row]);
    v[1] += (double)to_f32(elbo[row]);
    v[2] += (double)to_f32(nll[row]);
    if (kld) v[3] += (double)to_f32(kld[row]);
    if (add) v[4] += (double)to_f32(add[row]);
    if (mi) v[5] += (double)to_f32(mi[row]);
  }
  close_block_sums_d<6>(v, ws, inv_count, means, kNanElbo6);
}

template <typename T>
inline void launch_rows_means(const void* loss, const void* elbo, const void* nll, const void* kld, const void* add,
                              const void* mi, float* ws, float inv_count, float* means, int64_t B, hipStream_t stream,
                              int64_t steps = 1, void* loss_avg = nullptr, void* elbo_avg = nullptr,
                              void* elbo_lse = nullptr) {
  constexpr int VN = VecOf<T>::n;
  const bool vec = aligned16(loss) && aligned16(elbo) && aligned16(nll) && (!kld || aligned16(kld)) &&
                   (!add || aligned16(add)) && (!mi || aligned16(mi));
  const int64_t items = vec ? (B + VN - 1) / VN : B;
  int64_t blocks = (items + kBlock - 1) / kBlock;
  // small batches: ONE block (<= 8 vectors per thread and array) closes without the cross-block ticket chain
  const int grid = items <= 8 * kBlock ? 1 : (int)(blocks < 64 ? blocks : 64);
  if (vec && grid > 1 && blocks <= 64)   // one vector per thread: nothing to request ahead
    hipLaunchKernelGGL((elbo_rows_means_kernel<T, VN, 1>), dim3(grid), dim3(kBlock), 0, stream, (const T*)loss,
                       (const T*)elbo, (const T*)nll, (const T*)kld, (const T*)add, (const T*)mi, ws, inv_count, means, B,
                       steps, (T*)loss_avg, (T*)elbo_avg, (T*)elbo_lse);
  else if (vec)
    hipLaunchKernelGGL((elbo_rows_means_kernel<T, VN>), dim3(grid), dim3(kBlock), 0, stream, (const T*)loss, (const T*)elbo,
                       (const T*)nll, (const T*)kld, (const T*)add, (const T*)mi, ws, inv_count, means, B, steps,
                       (T*)loss_avg, (T*)elbo_avg, (T*)elbo_lse);
  else
    hipLaunchKernelGGL((elbo_rows_means_kernel<T, 1>), dim3(grid), dim3(kBlock), 0, stream, (const T*)loss, (const T*)elbo,
                       (const T*)nll, (const T*)kld, (const T*)add, (const T*)mi, ws, inv_count, means, B, steps,
                       (T*)loss_avg, (T*)elbo_avg, (T*)elbo_lse);
}

}  // namespace vae
