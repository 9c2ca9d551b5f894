// FlowReparameterizer "KL" reduction (the flow networks themselves are stock GEMMs).
//
// Replaces /root/reference reparameterizers/flow.py:67-78:
//   nll = (-0.5*u^2 - 0.5*log(2 pi)).sum(-1);  return mean(-nll - logdet, -1)
// 2-D logits: logdet is [B,1], so (-nll)[B] - logdet[B,1] broadcasts to [B,B] and the mean runs
// over the BATCH: kl_i = mean_j(nlp_j) - logdet_i, nlp_j = sum_f 0.5 u_jf^2 + F/2 log 2pi.
// The reference materialises the [B,B] matrix (17 GB at B = 65536); here it is O(B*F):
//   phase 1: nlp_j per row + block partial sums -> mean (one float; all-reduced when sharded)
//   phase 2: kl_i = mean - logdet_i
// 3-D logits [B,R,F], logdet [B,R]: kl_i = mean_r(nlp_ir - logdet_ir), a plain per-sample row
// over R*F values.
#include "common.hpp"
#include "../../include/vae_amd.h"

namespace vae {

constexpr float kHalfLog2PiF = 0.91893853320467274178f;

// rowval[row] = scale * sum_f 0.5 u^2 + cadd ; optional block partial sums of rowval
template <typename T, int V, int U>
__global__ __launch_bounds__(kBlock) void flow_rows_kernel(const T* __restrict__ u, T* __restrict__ rowval,
                                                          float* __restrict__ ws, float inv_count,
                                                          float* __restrict__ mean_out, int64_t B, int F, float scale,
                                                          float cadd, int G, int64_t tiles) {
  const int rpb = kBlock / G;
  const int sub = threadIdx.x % G;
  const int rloc = threadIdx.x / G;
  const int nvec = F / V;
  float acc[1] = {0.f};
  for (int64_t t = blockIdx.x; t < tiles; t += gridDim.x) {
    const int64_t row = t * rpb + rloc;
    float s = 0.f;
    if (row < B) {
      for (int j0 = sub; j0 < nvec; j0 += G * U) {
        float x[U][V];
        load_chunk<T, V, U>(u + row * F, j0, G, nvec, x);
#pragma unroll
        for (int k = 0; k < U; ++k)
          if (j0 + k * G < nvec)
#pragma unroll
            for (int i = 0; i < V; ++i) s += 0.5f * x[k][i] * x[k][i];
      }
    }
    s = group_sum(s, G);
    if (row < B && sub == 0) {
      const float v = to_f32(from_f32<T>(s * scale + cadd));
      rowval[row] = from_f32<T>(v);
      acc[0] += v;
    }
  }
  if (ws) grid_reduce<1>(acc, ws, inv_count, mean_out, 0x8u);
}

// 2-D: kl[i] = mean[0] - logdet[i].   3-D: kl[i] = rowval[i] - (1/R) sum_r logdet[i, r]
template <typename T>
__global__ __launch_bounds__(kBlock) void flow_combine_kernel(const T* __restrict__ logdet, const float* __restrict__ mean,
                                                             T* __restrict__ kl, int64_t B, int R, int cross_batch) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < B; i += (int64_t)gridDim.x * kBlock) {
    if (cross_batch) {
      kl[i] = from_f32<T>(mean[0] - to_f32(logdet[i]));
    } else {
      float s = 0.f;
      for (int r = 0; r < R; ++r) s += to_f32(logdet[i * R + r]);
      kl[i] = from_f32<T>(to_f32(kl[i]) - s / (float)R);
    }
  }
}

// gu[row, :] = coef(row) * u[row, :];  coef = gsum[0] * inv_count (2-D) or gkl[row / 1] / R (3-D rows are samples)
template <typename T, int V, int U>
__global__ __launch_bounds__(kBlock) void flow_bwd_kernel(const T* __restrict__ u, const T* __restrict__ gkl,
                                                         const float* __restrict__ gsum, T* __restrict__ gu,
                                                         int64_t B, int F, float scale, int G, int64_t tiles) {
  const int rpb = kBlock / G;
  const bool nts = nt_stores((int64_t)B * F * (int64_t)sizeof(T));
  const int sub = threadIdx.x % G;
  const int rloc = threadIdx.x / G;
  const int nvec = F / V;
  for (int64_t t = blockIdx.x; t < tiles; t += gridDim.x) {
    const int64_t row = t * rpb + rloc;
    if (row >= B) continue;
    const float c = (gsum ? gsum[0] : to_f32(gkl[row])) * scale;
    for (int j0 = sub; j0 < nvec; j0 += G * U) {
      float x[U][V];
      load_chunk<T, V, U>(u + row * F, j0, G, nvec, x);
#pragma unroll
      for (int k = 0; k < U; ++k) {
        const int j = j0 + k * G;
        if (j < nvec) {
#pragma unroll
          for (int i = 0; i < V; ++i) x[k][i] *= c;
          store_vec<T, V>(nts, gu + row * F + (int64_t)j * V, x[k]);
        }
      }
    }
  }
}


template <typename T>
static int flow_fwd_dispatch(const void* u, const void* logdet, int64_t R, void* kl, float* mean_io,
                             float* workspace, int64_t B, int64_t F, float inv_count, int cross_batch, int phase,
                             hipStream_t stream) {
  constexpr int VN = VecOf<T>::n;
  const int64_t row_len = cross_batch ? F : R * F;
  if (phase != 2) {
    const bool vec = (row_len % VN == 0) && aligned16(u);
    RowGeom g = row_geom(B, vec ? row_len / VN : row_len, kGridReduce);
    // load-only kernel: rows of 65..128 vectors run 16 rows per block, four chunks in flight (F = 512: 37.1 -> 29.3 us)
    if ((vec ? row_len / VN : row_len) > 64 && (vec ? row_len / VN : row_len) <= 128) regroup(g, B, 16, 4, kGridReduce);
    const float scale = cross_batch ? 1.0f : 1.0f / (float)R;
    const float cadd = (float)F * kHalfLog2PiF;
    float* partials = cross_batch ? workspace : nullptr;
    if (vec) {
      VAE_SWITCH_U(g.U, hipLaunchKernelGGL((flow_rows_kernel<T, VN, U>), dim3(g.grid), dim3(kBlock), 0, stream,
                                           (const T*)u, (T*)kl, partials, inv_count, mean_io, B, (int)row_len, scale, cadd, g.G,
                                           g.tiles));
    } else {
      VAE_SWITCH_U(g.U, hipLaunchKernelGGL((flow_rows_kernel<T, 1, U>), dim3(g.grid), dim3(kBlock), 0, stream,
                                           (const T*)u, (T*)kl, partials, inv_count, mean_io, B, (int)row_len, scale, cadd, g.G,
                                           g.tiles));
    }
  }
  if (phase != 1) {
    int64_t blocks = (B + kBlock - 1) / kBlock;
    const int grid = (int)(blocks < kReduceGridMax ? blocks : kReduceGridMax);
    hipLaunchKernelGGL((flow_combine_kernel<T>), dim3(grid), dim3(kBlock), 0, stream, (const T*)logdet, mean_io,
                       (T*)kl, B, (int)R, cross_batch);
  }
  return hipGetLastError() == hipSuccess ? VAE_OK : VAE_ERR_LAUNCH;
}

template <typename T>
static int flow_bwd_dispatch(const void* u, const void* gkl, const float* gsum, void* gu, int64_t B, int64_t row_len,
                             float scale, hipStream_t stream) {
  constexpr int VN = VecOf<T>::n;
  const bool vec = (row_len % VN == 0) && aligned16(u) && aligned16(gu);
  const RowGeom g = row_geom(B, vec ? row_len / VN : row_len, kGridStream);
  if (vec) {
    VAE_SWITCH_U(g.U, hipLaunchKernelGGL((flow_bwd_kernel<T, VN, U>), dim3(g.grid), dim3(kBlock), 0, stream,
                                         (const T*)u, (const T*)gkl, gsum, (T*)gu, B, (int)row_len, scale, g.G,
                                         g.tiles));
  } else {
    VAE_SWITCH_U(g.U, hipLaunchKernelGGL((flow_bwd_kernel<T, 1, U>), dim3(g.grid), dim3(kBlock), 0, stream,
                                         (const T*)u, (const T*)gkl, gsum, (T*)gu, B, (int)row_len, scale, g.G,
                                         g.tiles));
  }
  return hipGetLastError() == hipSuccess ? VAE_OK : VAE_ERR_LAUNCH;
}

}  // namespace vae

using namespace vae;

extern "C" {

int vae_flow_kl_fwd(const void* u, const void* logdet, int64_t R, void* kl, float* mean_io, float* workspace,
                    int64_t B, int64_t F, float inv_count, int cross_batch, int phase, int dtype,
                    vae_stream_t stream) {
  if (B < 0 || F <= 0 || R <= 0 || R * F > VAE_MAX_ROW || phase < 0 || phase > 2) return VAE_ERR_ARG;
  if (cross_batch && R != 1) return VAE_ERR_ARG;
  if (B == 0) return VAE_OK;
  if (!u || !kl || (phase != 1 && !logdet) || (cross_batch && (!mean_io || !workspace))) return VAE_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == VAE_F32)
    return flow_fwd_dispatch<float>(u, logdet, R, kl, mean_io, workspace, B, F, inv_count, cross_batch, phase, st);
  if (dtype == VAE_BF16)
    return flow_fwd_dispatch<bf16_t>(u, logdet, R, kl, mean_io, workspace, B, F, inv_count, cross_batch, phase, st);
  if (dtype == VAE_F16)
    return flow_fwd_dispatch<f16_t>(u, logdet, R, kl, mean_io, workspace, B, F, inv_count, cross_batch, phase, st);
  return VAE_ERR_ARG;
}

int vae_flow_kl_bwd(const void* u, const void* gkl, const float* gsum, void* gu, int64_t B, int64_t row_len,
                    float scale, int dtype, vae_stream_t stream) {
  if (B < 0 || row_len <= 0 || row_len > VAE_MAX_ROW) return VAE_ERR_ARG;
  if (B == 0) return VAE_OK;
  if (!u || !gu || (!gkl && !gsum)) return VAE_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == VAE_F32) return flow_bwd_dispatch<float>(u, gkl, gsum, gu, B, row_len, scale, st);
  if (dtype == VAE_BF16) return flow_bwd_dispatch<bf16_t>(u, gkl, gsum, gu, B, row_len, scale, st);
  if (dtype == VAE_F16) return flow_bwd_dispatch<f16_t>(u, gkl, gsum, gu, B, row_len, scale, st);
  return VAE_ERR_ARG;
}

}  // extern "C"
