// Streaming-ceiling probe for the hot path's byte mixes (tuning aid, not part of the library):
//   hipcc -O3 --offload-arch=gfx950 scripts/membench.hip -o scripts/membench && scripts/membench
// Flat 16-B-per-lane kernels with R read streams and W write streams over buffers that rotate through
// NSETS > 256 MiB of distinct memory, so the Infinity Cache never holds a pass's data (what the kernels of a
// training step see).  Prints TB/s per variant: the rate a row kernel with the same mix can hope for.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

template <bool NT> __device__ __forceinline__ f4 ld(const f4* p) {
  if constexpr (NT) return __builtin_nontemporal_load(p); else return *p;
}
template <bool NT> __device__ __forceinline__ void st(f4* p, f4 v) {
  if constexpr (NT) __builtin_nontemporal_store(v, p); else *p = v;
}

// each block: U consecutive 4-KiB chunks per stream (256 lanes x 16 B), all loads first
template <int R, int W, int U, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void stream_kernel(const f4* __restrict__ r0, const f4* __restrict__ r1,
                                                     const f4* __restrict__ r2, const f4* __restrict__ r3,
                                                     f4* __restrict__ w0, f4* __restrict__ w1, size_t nvec) {
  const f4* rs[4] = {r0, r1, r2, r3};
  f4* ws[2] = {w0, w1};
  for (size_t base = (size_t)blockIdx.x * 256 * U; base < nvec; base += (size_t)gridDim.x * 256 * U) {
    f4 v[R][U];
#pragma unroll
    for (int s = 0; s < R; ++s)
#pragma unroll
      for (int u = 0; u < U; ++u) {
        size_t i = base + (size_t)u * 256 + threadIdx.x;
        i = i < nvec ? i : nvec - 1;
        v[s][u] = ld<NTL>(rs[s] + i);
      }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t i = base + (size_t)u * 256 + threadIdx.x;
      f4 a = v[0][u];
#pragma unroll
      for (int s = 1; s < R; ++s) a = a * 1.0001f + v[s][u];
      if (i < nvec) {
#pragma unroll
        for (int s = 0; s < W; ++s) st<NTS>(ws[s] + i, a + (float)s);
      }
    }
  }
}

// the library's row geometry: one wave per row of `rvec` 16-B vectors, 4 rows per block, U = 4 loads per lane and
// stream issued up front (indices past the row end clamped), one tile per block
template <int R, int W, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void row_kernel(const f4* __restrict__ r0, const f4* __restrict__ r1,
                                                  const f4* __restrict__ r2, f4* __restrict__ w0, size_t rows, int rvec) {
  const f4* rs[3] = {r0, r1, r2};
  const int lane = threadIdx.x & 63;
  const size_t row = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  for (int j0 = lane; j0 < rvec; j0 += 256) {
    f4 v[R][4];
#pragma unroll
    for (int s = 0; s < R; ++s)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        int j = j0 + u * 64;
        j = j < rvec ? j : rvec - 1;
        v[s][u] = ld<NTL>(rs[s] + row * rvec + j);
      }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int j = j0 + u * 64;
      f4 a = v[0][u];
#pragma unroll
      for (int s = 1; s < R; ++s) a = a * 1.0001f + v[s][u];
      if (W && j < rvec) st<NTS>(w0 + row * rvec + j, a);
    }
  }
}

// short rows (16-bit tensors: 784 px = 98 vectors): G lanes per row, RW rows per lane group in flight, UU chunks per row
template <int G, int RW, int UU, bool NTS>
__global__ __launch_bounds__(256) void short_row_kernel(const f4* __restrict__ r0, const f4* __restrict__ r1,
                                                        f4* __restrict__ w0, size_t rows, int rvec) {
  constexpr int RPB = 256 / G;
  const int sub = threadIdx.x % G, rloc = threadIdx.x / G;
  f4 a[RW][UU], b[RW][UU];
#pragma unroll
  for (int q = 0; q < RW; ++q) {
    size_t row = ((size_t)blockIdx.x * RW + q) * RPB + rloc;
    row = row < rows ? row : rows - 1;
#pragma unroll
    for (int u = 0; u < UU; ++u) {
      int j = sub + u * G;
      j = j < rvec ? j : rvec - 1;
      a[q][u] = ld<true>(r0 + row * rvec + j);
      b[q][u] = ld<true>(r1 + row * rvec + j);
    }
  }
#pragma unroll
  for (int q = 0; q < RW; ++q) {
    const size_t row = ((size_t)blockIdx.x * RW + q) * RPB + rloc;
#pragma unroll
    for (int u = 0; u < UU; ++u) {
      const int j = sub + u * G;
      if (row < rows && j < rvec) st<NTS>(w0 + row * rvec + j, a[q][u] * 1.0001f + b[q][u]);
    }
  }
}

template <int G, int RW, int UU, bool NTS>
double run_short(const char* name, std::vector<f4*>& pool, size_t nvec, int rvec, int iters) {
  const int nsets = (int)pool.size() / 3;
  const size_t rows = nvec / rvec;
  const size_t per_block = (size_t)(256 / G) * RW;
  const size_t blocks = (rows + per_block - 1) / per_block;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto launch = [&](int it) {
    f4** b = &pool[(it % nsets) * 3];
    hipLaunchKernelGGL((short_row_kernel<G, RW, UU, NTS>), dim3((unsigned)blocks), dim3(256), 0, 0, b[0], b[1], b[2], rows,
                       rvec);
  };
  for (int i = 0; i < 3; ++i) launch(i);
  std::vector<double> t;
  for (int rep = 0; rep < 5; ++rep) {
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) launch(i);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    t.push_back(ms / iters);
  }
  std::sort(t.begin(), t.end());
  const double bytes = (double)rows * rvec * 16 * 3;
  const double tbs = bytes / (t[2] * 1e-3) / 1e12;
  printf("%-34s 2R1W rows of %d B G%d x%d rows U%d nts%d grid %-7zu  %8.1f us  %5.2f TB/s (%.3f of 8)\n", name, rvec * 16, G,
         RW, UU, (int)NTS, blocks, t[2] * 1e3, tbs, tbs / 8.0);
  fflush(stdout);
  return tbs;
}

template <int R, int W, bool NTL, bool NTS>
double run_rows(const char* name, std::vector<f4*>& pool, size_t nvec, int rvec, int iters) {
  const int per = R + W;
  const int nsets = (int)pool.size() / per;
  const size_t rows = nvec / rvec;
  const size_t blocks = (rows + 3) / 4;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto launch = [&](int it) {
    f4** b = &pool[(it % nsets) * per];
    hipLaunchKernelGGL((row_kernel<R, W, NTL, NTS>), dim3((unsigned)blocks), dim3(256), 0, 0, b[0], R > 1 ? b[1] : nullptr,
                       R > 2 ? b[2] : nullptr, b[R], rows, rvec);
  };
  for (int i = 0; i < 3; ++i) launch(i);
  std::vector<double> t;
  for (int rep = 0; rep < 5; ++rep) {
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) launch(i);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    t.push_back(ms / iters);
  }
  std::sort(t.begin(), t.end());
  const double bytes = (double)rows * rvec * 16 * per;
  const double tbs = bytes / (t[2] * 1e-3) / 1e12;
  printf("%-34s R%d W%d rows of %d B ntl%d nts%d grid %-7zu  %8.1f us  %5.2f TB/s (%.3f of 8)\n", name, R, W, rvec * 16,
         (int)NTL, (int)NTS, blocks, t[2] * 1e3, tbs, tbs / 8.0);
  fflush(stdout);
  return tbs;
}


// persistent forms of row_kernel (2R1W): `grid` blocks walk the tiles grid-strided.  PIPE = 0: loads -> compute ->
// stores per tile (a wave's next loads are younger than its stores, and vmcnt retires in order: every tile waits for
// the acknowledgement of the previous tile's stores).  PIPE = 1: the next tile's loads are issued BEFORE this tile's
// stores.
template <bool NTS, int PIPE>
__global__ __launch_bounds__(256) void row_persist_kernel(const f4* __restrict__ r0, const f4* __restrict__ r1,
                                                          f4* __restrict__ w0, size_t rows, int rvec) {
  const int lane = threadIdx.x & 63;
  const size_t ntiles = (rows + 3) / 4;
  auto ldrow = [&](size_t row, f4 (&a)[4], f4 (&b)[4]) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      int j = lane + u * 64;
      j = j < rvec ? j : rvec - 1;
      a[u] = ld<true>(r0 + row * rvec + j);
      b[u] = ld<true>(r1 + row * rvec + j);
    }
  };
  auto strow = [&](size_t row, const f4 (&a)[4], const f4 (&b)[4]) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int j = lane + u * 64;
      if (j < rvec) st<NTS>(w0 + row * rvec + j, a[u] * 1.0001f + b[u]);
    }
  };
  size_t t = blockIdx.x;
  if (t >= ntiles) return;
  if (PIPE == 0) {
    for (; t < ntiles; t += gridDim.x) {
      const size_t row = t * 4 + (threadIdx.x >> 6);
      if (row >= rows) continue;
      f4 a[4], b[4];
      ldrow(row, a, b);
      strow(row, a, b);
    }
  } else {
    f4 a0[4], b0[4], a1[4], b1[4];
    size_t row = t * 4 + (threadIdx.x >> 6);
    if (row >= rows) row = rows - 1;
    ldrow(row, a0, b0);
    for (;;) {
      size_t tn = t + gridDim.x;
      size_t rown = tn * 4 + (threadIdx.x >> 6);
      const bool more = tn < ntiles;
      if (more) ldrow(rown < rows ? rown : rows - 1, a1, b1);
      if (t * 4 + (threadIdx.x >> 6) < rows) strow(t * 4 + (threadIdx.x >> 6), a0, b0);
      if (!more) break;
      t = tn;
      tn = t + gridDim.x;
      rown = tn * 4 + (threadIdx.x >> 6);
      const bool more2 = tn < ntiles;
      if (more2) ldrow(rown < rows ? rown : rows - 1, a0, b0);
      if (t * 4 + (threadIdx.x >> 6) < rows) strow(t * 4 + (threadIdx.x >> 6), a1, b1);
      if (!more2) break;
      t = tn;
    }
  }
}

template <bool NTS, int PIPE>
double run_persist(const char* name, std::vector<f4*>& pool, size_t nvec, int rvec, int grid, int iters) {
  const int per = 3;
  const int nsets = (int)pool.size() / per;
  const size_t rows = nvec / rvec;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto launch = [&](int it) {
    f4** b = &pool[(it % nsets) * per];
    hipLaunchKernelGGL((row_persist_kernel<NTS, PIPE>), dim3(grid), dim3(256), 0, 0, b[0], b[1], b[2], rows, rvec);
  };
  for (int i = 0; i < 3; ++i) launch(i);
  std::vector<double> t;
  for (int rep = 0; rep < 5; ++rep) {
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) launch(i);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    t.push_back(ms / iters);
  }
  std::sort(t.begin(), t.end());
  const double bytes = (double)rows * rvec * 16 * per;
  const double tbs = bytes / (t[2] * 1e-3) / 1e12;
  printf("%-34s 2R1W rows of %d B nts%d pipe%d grid %-6d  %8.1f us  %5.2f TB/s (%.3f of 8)\n", name, rvec * 16, (int)NTS, PIPE,
         grid, t[2] * 1e3, tbs, tbs / 8.0);
  fflush(stdout);
  return tbs;
}

template <int R, int W, int U, bool NTL, bool NTS>
double run(const char* name, std::vector<f4*>& pool, size_t nvec, long grid_cap, int iters) {
  const int per = R + W;
  const int nsets = (int)pool.size() / per;
  size_t blocks = (nvec + 256 * U - 1) / (256 * U);
  if (grid_cap > 0 && (long)blocks > grid_cap) blocks = grid_cap;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto launch = [&](int it) {
    f4** b = &pool[(it % nsets) * per];
    const f4* r[4] = {b[0], R > 1 ? b[1] : nullptr, R > 2 ? b[2] : nullptr, R > 3 ? b[3] : nullptr};
    f4* w[2] = {W > 0 ? b[R] : nullptr, W > 1 ? b[R + 1] : nullptr};
    hipLaunchKernelGGL((stream_kernel<R, W, U, NTL, NTS>), dim3((unsigned)blocks), dim3(256), 0, 0, r[0], r[1], r[2], r[3], w[0], w[1], nvec);
  };
  for (int i = 0; i < 3; ++i) launch(i);
  std::vector<double> t;
  for (int rep = 0; rep < 5; ++rep) {
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) launch(i);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    t.push_back(ms / iters);
  }
  std::sort(t.begin(), t.end());
  const double bytes = (double)nvec * 16 * per;
  const double tbs = bytes / (t[2] * 1e-3) / 1e12;
  printf("%-34s R%d W%d U%d ntl%d nts%d grid %-7zu  %8.1f us  %5.2f TB/s (%.3f of 8)\n", name, R, W, U, (int)NTL, (int)NTS,
         blocks, t[2] * 1e3, tbs, tbs / 8.0);
  fflush(stdout);
  return tbs;
}

int main(int argc, char** argv) {
  const size_t mb = argc > 1 ? atol(argv[1]) : 205;     // MB per stream (C2's x / recon / grecon: 205.5 MB)
  const size_t nvec = mb * 1000 * 1000 / 16;
  const int nbuf = argc > 2 ? atoi(argv[2]) : 18;        // distinct buffers in the pool (18 x 205 MB = 3.7 GB)
  std::vector<f4*> pool(nbuf);
  for (auto& p : pool) { CK(hipMalloc(&p, nvec * 16)); CK(hipMemset(p, 0, nvec * 16)); }
  CK(hipDeviceSynchronize());
  const int it = 18;
  // the step's mixes: NLL one-pass 2R1W, gauss fwd 3R1W, gauss bwd 4R2W; plus pure read / write / copy
  run_rows<2, 1, true, false>("rows 2R1W 784 px", pool, nvec, 196, it);
  run_rows<2, 1, true, true>("rows 2R1W 784 px nt stores", pool, nvec, 196, it);
  run_rows<2, 1, true, false>("rows 2R1W 768 px", pool, nvec, 192, it);
  run_rows<2, 1, true, false>("rows 2R1W 1024 px", pool, nvec, 256, it);
  run_rows<2, 1, true, false>("rows 2R1W 800 px", pool, nvec, 200, it);
  run_rows<3, 1, true, false>("rows 3R1W 512", pool, nvec, 128, it);
  run_rows<3, 1, true, true>("rows 3R1W 512 nt stores", pool, nvec, 128, it);
  if (argc > 3 && argv[3][0] == 'w') {   // write-heavy mixes (r03): the long-row NLL backward is 3R2W (mb = 402, nbuf = 10)
    run<3, 2, 4, true, true>("3R2W nt stores", pool, nvec, 0, it);
    run<3, 2, 4, true, false>("3R2W plain stores", pool, nvec, 0, it);
    run<3, 2, 2, true, true>("3R2W nt stores U2", pool, nvec, 0, it);
    run<3, 2, 1, true, true>("3R2W nt stores U1", pool, nvec, 0, it);
    run<3, 1, 4, true, true>("3R1W nt stores", pool, nvec, 0, it);
    run<3, 0, 4, true, true>("3R0W", pool, nvec, 0, it);
    run<1, 2, 4, true, true>("1R2W nt stores", pool, nvec, 0, it);
    run<2, 2, 4, true, true>("2R2W nt stores", pool, nvec, 0, it);
    run<4, 2, 4, true, true>("4R2W nt stores", pool, nvec, 0, it);
    return 0;
  }
  if (argc > 3 && argv[3][0] == 's') {   // short rows: mb = 103 for the bf16 C2 streams
    run_short<64, 1, 2, true>("short G64 1 row U2 (library)", pool, nvec, 98, it);
    run_short<64, 2, 2, true>("short G64 2 rows U2", pool, nvec, 98, it);
    run_short<64, 4, 2, true>("short G64 4 rows U2", pool, nvec, 98, it);
    run_short<32, 1, 4, true>("short G32 1 row U4", pool, nvec, 98, it);
    run_short<32, 2, 4, true>("short G32 2 rows U4", pool, nvec, 98, it);
    run_short<16, 1, 7, true>("short G16 1 row U7", pool, nvec, 98, it);
    run_short<16, 2, 7, true>("short G16 2 rows U7", pool, nvec, 98, it);
    run_short<64, 1, 2, false>("short G64 1 row U2 plain st", pool, nvec, 98, it);
    run_short<64, 1, 2, true>("short 128-vector rows G64 U2", pool, nvec, 128, it);
    run_short<64, 2, 2, true>("short 128-vector rows G64 x2", pool, nvec, 128, it);
    return 0;
  }
  if (argc > 3) {
    for (int grid : {1024, 2048, 4096}) {
      run_persist<true, 0>("persistent naive", pool, nvec, 196, grid, it);
      run_persist<true, 1>("persistent pipelined", pool, nvec, 196, grid, it);
    }
    run_persist<false, 0>("persistent naive plain st", pool, nvec, 196, 2048, it);
    run_persist<false, 1>("persistent pipelined plain st", pool, nvec, 196, 2048, it);
    return 0;
  }
  run<0 + 1, 1, 4, true, false>("copy 1R1W", pool, nvec, 0, it);
  run<2, 1, 4, true, false>("2R1W one-tile blocks", pool, nvec, 0, it);
  run<2, 1, 4, false, false>("2R1W plain loads", pool, nvec, 0, it);
  run<2, 1, 4, true, true>("2R1W nt stores", pool, nvec, 0, it);
  run<2, 1, 2, true, false>("2R1W U2", pool, nvec, 0, it);
  run<2, 1, 1, true, false>("2R1W U1", pool, nvec, 0, it);
  run<2, 1, 8, true, false>("2R1W U8", pool, nvec, 0, it);
  run<2, 1, 4, true, false>("2R1W persistent 2048", pool, nvec, 2048, it);
  run<2, 1, 4, true, false>("2R1W persistent 1024", pool, nvec, 1024, it);
  run<2, 1, 4, true, false>("2R1W persistent 4096", pool, nvec, 4096, it);
  run<3, 1, 4, true, false>("3R1W", pool, nvec, 0, it);
  run<3, 1, 2, true, false>("3R1W U2", pool, nvec, 0, it);
  run<4, 2, 2, true, false>("4R2W U2", pool, nvec, 0, it);
  run<4, 2, 4, true, false>("4R2W U4", pool, nvec, 0, it);
  run<4, 1, 2, true, false>("4R1W U2", pool, nvec, 0, it);
  return 0;
}
