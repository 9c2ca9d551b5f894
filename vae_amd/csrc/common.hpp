// Shared device helpers for the vae_amd HIP kernels (gfx950 / CDNA4, wave64).
//
// Every kernel on this path is a streaming row kernel: a "row" is one sample's
// feature vector (L latents, K categories or D pixels).  A row is owned by a
// group of G consecutive lanes (G a power of two, 1..64, picked on the host from
// the row length), each lane moves 16 bytes per access (4 x f32 or 8 x bf16) so
// a wave issues 1 KiB coalesced requests, row reductions are wave64 shuffles
// inside the group, and the few batch-level scalars (means) leave the kernel as
// one partial per block, summed in a fixed order by the block that arrives last
// (grid_reduce / grid_reduce_cluster below; deterministic: no float atomics).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

namespace vae {

constexpr int kBlock = 256;     // 4 waves per workgroup

struct bf16_t {
  uint16_t bits;
};
struct f16_t {  // IEEE binary16: what the reference's `half=True` configs run in (helpers.utils.same_type)
  uint16_t bits;
};
template <typename T>
struct DTypeCode;
template <>
struct DTypeCode<float> {
  static constexpr int value = 0;
};
template <>
struct DTypeCode<bf16_t> {
  static constexpr int value = 1;
};
template <>
struct DTypeCode<f16_t> {
  static constexpr int value = 2;
};

__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v.bits) << 16); }
__device__ __forceinline__ float to_f32(f16_t v) { return (float)__builtin_bit_cast(_Float16, v.bits); }

template <typename T>
__device__ __forceinline__ T from_f32(float v);
template <>
__device__ __forceinline__ float from_f32<float>(float v) {
  return v;
}
template <>
__device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) {
  // plain cast: hipcc emits v_cvt_pk_bf16_f32 (round-nearest-even, NaN stays NaN)
  __bf16 h = static_cast<__bf16>(v);
  bf16_t r;
  r.bits = __builtin_bit_cast(uint16_t, h);
  return r;
}
template <>
__device__ __forceinline__ f16_t from_f32<f16_t>(float v) {
  _Float16 h = static_cast<_Float16>(v);  // v_cvt_f16_f32: round-nearest-even, overflow to inf
  f16_t r;
  r.bits = __builtin_bit_cast(uint16_t, h);
  return r;
}

// Individually rounded ops.  hipcc's default -ffp-contract=fast turns a*b+c into one fma
// (and HIP's __fmul_rn/__fadd_rn are plain operators, so they fuse too); where the
// reference's op sequence must be reproduced bit for bit (z, logvar, the Gumbel logits)
// the product and the sum have to round separately, as two torch ops do.
__device__ __forceinline__ float mul_rn(float a, float b) {
#pragma clang fp contract(off)
  return a * b;
}
__device__ __forceinline__ float add_rn(float a, float b) {
#pragma clang fp contract(off)
  return a + b;
}
__device__ __forceinline__ float sub_rn(float a, float b) {
#pragma clang fp contract(off)
  return a - b;
}
__device__ __forceinline__ float div_rn(float a, float b) {
#pragma clang fp contract(off)
  return a / b;
}

// 16-byte packets: the widest per-lane access (global_load/store_dwordx4).  Every tensor on
// this path is streamed exactly once per kernel, so loads carry the non-temporal hint (no reuse
// to keep in L2 / Infinity Cache): +8..15 % on the gauss kernels (r01, scripts/ab.py).  Stores are
// non-temporal only when the launch is too large to be cache-resident (one [B, row] tensor of at
// least kNtStoreMinBytes): measured r02 in the step sequence of bench.py, where every kernel
// finds the caches cold, nt stores take 6 % off the whole step (gauss_bwd 135 -> 125 us, the
// one-pass NLL 123 -> 114 us; scripts/membench.hip: 2R1W rows of 3136 B 112 -> 100 us), while on
// small tensors that the next kernel reads back from L2 / Infinity Cache they cost up to 60 %
// (gumbel K = 32).  Round 1 had measured nt stores as a loss because it looped each kernel alone
// on a working set that the 256 MiB Infinity Cache partly held.
template <typename T, int V>
struct PackOf;
template <>
struct PackOf<float, 4> {
  typedef float type __attribute__((ext_vector_type(4)));
};
template <>
struct PackOf<bf16_t, 8> {
  typedef uint32_t type __attribute__((ext_vector_type(4)));
};
template <>
struct PackOf<f16_t, 8> {
  typedef uint32_t type __attribute__((ext_vector_type(4)));
};

// Two floats -> one dword of 16-bit values, low half first.  Converted as a PAIR: one v_cvt_pk_bf16_f32 (two
// v_cvt_f16_f32 + v_pack_b32_f16) instead of a conversion per element, a shift and an or -- 2 VALU slots per stored
// element less in every 16-bit kernel (r03: those are VALU-co-bound).  Same rounding as from_f32 (nearest-even).
template <typename T>
__device__ __forceinline__ uint32_t pack2(float a, float b);
template <>
__device__ __forceinline__ uint32_t pack2<bf16_t>(float a, float b) {
  typedef float f2 __attribute__((ext_vector_type(2)));
  typedef __bf16 h2 __attribute__((ext_vector_type(2)));
  const f2 v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, h2));
}
template <>
__device__ __forceinline__ uint32_t pack2<f16_t>(float a, float b) {
  typedef float f2 __attribute__((ext_vector_type(2)));
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  const f2 v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, h2));
}

template <typename T>
struct VecOf {
  static constexpr int n = 16 / sizeof(T);
};

#ifndef VAE_NT_LOAD
#define VAE_NT_LOAD 1
#endif
#ifndef VAE_NT_STORE_MIN_BYTES
#define VAE_NT_STORE_MIN_BYTES (32ll << 20)
#endif
constexpr long long kNtStoreMinBytes = VAE_NT_STORE_MIN_BYTES;
__device__ __forceinline__ bool nt_stores(int64_t tensor_bytes) { return tensor_bytes >= kNtStoreMinBytes; }

template <typename T, int V, bool NT = (VAE_NT_LOAD != 0)>
__device__ __forceinline__ void load_vec(const T* __restrict__ p, float (&out)[V]) {
  if constexpr (V == 1) {
    out[0] = to_f32(p[0]);
  } else {
    typedef typename PackOf<T, V>::type P;
    P raw;
    if constexpr (NT) {
      raw = __builtin_nontemporal_load(reinterpret_cast<const P*>(p));
    } else {
      raw = *reinterpret_cast<const P*>(p);
    }
    if constexpr (sizeof(T) == 4) {
#pragma unroll
      for (int i = 0; i < V; ++i) out[i] = raw[i];
    } else if constexpr (DTypeCode<T>::value == 2) {
#pragma unroll
      for (int i = 0; i < V / 2; ++i) {
        out[2 * i] = (float)__builtin_bit_cast(_Float16, (uint16_t)(raw[i] & 0xffffu));
        out[2 * i + 1] = (float)__builtin_bit_cast(_Float16, (uint16_t)(raw[i] >> 16));
      }
    } else {
#pragma unroll
      for (int i = 0; i < V / 2; ++i) {
        out[2 * i] = __uint_as_float(raw[i] << 16);
        out[2 * i + 1] = __uint_as_float(raw[i] & 0xffff0000u);
      }
    }
  }
}

template <typename T, int V>
__device__ __forceinline__ void store_vec(bool nts, T* __restrict__ p, const float (&in)[V]) {
  if constexpr (V == 1) {
    p[0] = from_f32<T>(in[0]);
  } else {
    typedef typename PackOf<T, V>::type P;
    P raw;
    if constexpr (sizeof(T) == 4) {
#pragma unroll
      for (int i = 0; i < V; ++i) raw[i] = in[i];
    } else {
#pragma unroll
      for (int i = 0; i < V / 2; ++i) raw[i] = pack2<T>(in[2 * i], in[2 * i + 1]);
    }
    if (nts) {   // uniform over the launch (a function of the kernel arguments): a scalar branch
      // Spelled as the instruction: with __builtin_nontemporal_store here the two arms are merged into one plain
      // store whatever is put between them (the hint is metadata, dropped when the stores are sunk into the join
      // block).  The compiler does not count this store in its vmcnt bookkeeping, so a later counted wait may also
      // wait for its acknowledgement; measured against a build with the builtin in every store (r02, step sequence):
      // no difference, every load of a tile is issued before its first store.
      // The s_nop is the wait state the hardware needs between a store of more than 64 bits and a VALU write of its
      // data registers: the compiler pads its own stores, it cannot see this one (without it bernoulli_bwd stored
      // values of the next chunk: tests/test_gpu_bigindex.py).  The hazard, by name: "VMEM store of more than 8 bytes
      // of data followed by a VALU write of the VGPRs that hold the write data" (CDNA3 / CDNA4 ISA guide, section
      // "Manually Inserted Wait States (NOPs)"; LLVM GCNHazardRecognizer::checkVALUHazards -> createsVALUHazard), 1
      // wait state on gfx90a, 2 on gfx940 and later: `s_nop 1` is those two.  tests/test_gpu_ntstore.py drives every
      // kernel family through this branch against the oracle.
      static_assert(sizeof(P) == 16, "the wait states below are for a 128-bit store");
      asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" ::"v"(reinterpret_cast<P*>(p)), "v"(raw) : "memory");
    } else {
      *reinterpret_cast<P*>(p) = raw;
    }
  }
}
template <typename T, int V>
__device__ __forceinline__ void store_vec(T* __restrict__ p, const float (&in)[V]) {
  store_vec<T, V>(false, p, in);
}

// U row-chunks of one lane, all loads issued before any use: lane `sub` of a G-lane group owns
// vectors j0, j0+G, ... of the row.  Indices past the row end are clamped to the last vector
// (always a valid address, an L1 hit, masked by the caller with `j < nvec`), so the loads carry
// no branch and the compiler batches them under one counted s_waitcnt: 16*U bytes per tensor
// in flight per lane, which is what keeps HBM busy at 8 waves/SIMD.
template <typename T, int V, int U, bool NT = (VAE_NT_LOAD != 0)>
__device__ __forceinline__ void load_chunk(const T* __restrict__ base, int j0, int G, int nvec, float (&out)[U][V]) {
#pragma unroll
  for (int u = 0; u < U; ++u) {
    int j = j0 + u * G;
    j = j < nvec ? j : nvec - 1;
    load_vec<T, V, NT>(base + (int64_t)j * V, out[u]);
  }
}

// cheap transcendental forms (v_exp_f32 / v_log_f32 / v_rcp_f32, ~1-2 ulp): used where the
// result feeds a row sum or a gradient whose tolerance is rtol 1e-5, never where a value has
// to match the reference bit for bit.
__device__ __forceinline__ float fast_rcp(float v) { return __builtin_amdgcn_rcpf(v); }
// 1/d to ~0.5 ulp: v_rcp_f32 + one Newton step (3 instructions instead of the ~10 of an IEEE divide)
__device__ __forceinline__ float rcp_nr(float d) {
  const float r = __builtin_amdgcn_rcpf(d);
  return mul_rn(r, __builtin_fmaf(-d, r, 2.0f));   // spelled out: the same bits in every kernel that uses it
}
// ln(x) for x in the NORMAL float range: bare v_log_f32 (log2, 1 ulp) times ln 2.  hipcc's __logf wraps the
// same instruction in denormal scaling and a compensated multiply (~12 VALU ops instead of 2), which made the
// Bernoulli NLL forward issue-bound (672 G elements/s).  Callers guarantee x >= ~1e-30 or accept -inf at 0.
constexpr float kLn2 = 0.693147180559945309f;
constexpr float kLog2e = 1.44269504088896341f;
__device__ __forceinline__ float log2_raw(float x) { return __builtin_amdgcn_logf(x); }
__device__ __forceinline__ float fast_log(float x) { return kLn2 * __builtin_amdgcn_logf(x); }
// e^x for x <= ~80 with a result in the normal range or flushing to 0 (bare v_exp_f32)
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * kLog2e); }
// e^x to ~1 ulp: the product x*log2(e) is split into its rounded value and the residual (fma) plus the low word of
// log2(e), v_exp_f32 takes the rounded part and the residual is applied as a first-order correction -- 8 VALU ops
// against the 14 of ocml's expf (no ldexp / range selects).  Beyond |x| = 88 (and for +-inf, whose residual is NaN,
// and NaN) the bare v_exp_f32 value is returned: inf / 0 / huge / NaN as they should be, where a last-ulp correction
// means nothing.  Every product / fma is spelled out, so the bits are the same in every kernel that uses it.
__device__ __forceinline__ float exp_acc(float x) {
  const float t = mul_rn(x, 1.44269502f);
  const float r = __builtin_fmaf(x, 1.92596299e-8f, __builtin_fmaf(x, 1.44269502f, -t));
  const float e = __builtin_amdgcn_exp2f(t);
  const float c = __builtin_fmaf(e, mul_rn(r, kLn2), e);
  return fabsf(x) < 88.0f ? c : e;
}
// log(1 + e) for e in [0, 1]
__device__ __forceinline__ float log1p_unit(float e) {
  return e < 1e-3f ? e * (1.0f - e * (0.5f - e * (1.0f / 3.0f))) : fast_log(1.0f + e);
}

// log(1 + e) for any e >= 0 with full RELATIVE precision (Kahan: log(u) * e / (u - 1), u = fl(1 + e));
// plain log(1 + e) loses eps/e of it, which a Normal scale softplus(v) + 1e-6 then squares.
__device__ __forceinline__ float log1p_unit_any(float e) {
  const float u = 1.0f + e;
  const float d = u - 1.0f;
  return d == 0.0f ? e : fast_log(u) * (e * fast_rcp(d));
}

// Workgroup barrier for LDS traffic only.  __syncthreads() is a workgroup-scope fence as well: it puts
// `s_waitcnt vmcnt(0)` in front of s_barrier, so every wave first waits until its global stores are acknowledged
// -- 1.5-4 us for a wave that has just issued its row of output (r02: the block-level partial sums of the one-pass
// NLL cost 4.7 us per launch with it, 1.1 us with nothing stored in front).  Nothing on this path hands global
// memory from one wave of a block to another, so only the LDS writes have to be complete.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// One DPP step (v_mov_b32_dpp: a cross-lane read inside a 16-lane row at VALU speed, ~10 cycles, where a ds_bpermute
// butterfly step is an LDS round trip of ~120).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false));
}

// The 16- and 32-lane levels of the butterfly: gfx950's v_permlane16_swap / v_permlane32_swap exchange the odd 16-lane
// rows of one register with the even rows of another (the upper and lower 32 lanes for the second), at VALU speed.
// Called with two copies of v, the first register comes back holding the even-row (lower-half) values in both rows
// of every pair and the second the odd-row (upper-half) values: their sum / max IS the xor-16 / xor-32 butterfly step,
// the same bits in both partners -- where ds_bpermute (__shfl_xor) is an LDS round trip of ~120 cycles behind an
// lgkmcnt wait.  Inline asm: the builtin's two results are mis-selected by hipcc 7.2 (both read the first register:
// scripts/exp/permlane_test.hip checks this form against __shfl_xor on the GPU); the s_nop is the wait state the
// instruction needs behind the VALU writes of its operands, which the compiler cannot see here.
#ifndef VAE_NO_PERMLANE_SWAP
__device__ __forceinline__ void swap16(float v, float& a, float& b) {
  a = v; b = v;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void swap32(float v, float& a, float& b) {
  a = v; b = v;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
#else
__device__ __forceinline__ void swap16(float v, float& a, float& b) { a = v; b = __shfl_xor(v, 16, 64); }
__device__ __forceinline__ void swap32(float v, float& a, float& b) { a = v; b = __shfl_xor(v, 32, 64); }
#endif

// Sum / max over the G lanes that share a row (G power of two, <= 64; every lane of the wave takes part: the call
// sites are in block-uniform control flow).  Butterfly: xor 1 and xor 2 inside the quads, then the half-row and row
// mirrors (the sub-groups they pair already hold equal values) by DPP; the 16- and 32-lane levels by permlane swaps.
// After it every lane of the group holds the result, the same bits in all of them.
__device__ __forceinline__ float group_sum(float v, int G) {
  if (G >= 2) v += dpp_f<0xB1, 0xf>(v);    // quad_perm [1,0,3,2]
  if (G >= 4) v += dpp_f<0x4E, 0xf>(v);    // quad_perm [2,3,0,1]
  if (G >= 8) v += dpp_f<0x141, 0xf>(v);   // row_half_mirror
  if (G >= 16) v += dpp_f<0x140, 0xf>(v);  // row_mirror
  float a, b;
  if (G >= 32) { swap16(v, a, b); v = a + b; }
  if (G >= 64) { swap32(v, a, b); v = a + b; }
  return v;
}
// max(a, b) as the bare instruction.  fmaxf() costs three: the compiler quiets both operands first (v_max_f32 x, x)
// because IEEE maxNum must not return a signalling NaN -- which v_max_f32 in the IEEE mode these kernels run in does
// not do either.  Same result as fmaxf for every input, a NaN operand included (the other one is returned).
__device__ __forceinline__ float max_raw(float a, float b) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float group_max(float v, int G) {
  if (G >= 2) v = max_raw(v, dpp_f<0xB1, 0xf>(v));
  if (G >= 4) v = max_raw(v, dpp_f<0x4E, 0xf>(v));
  if (G >= 8) v = max_raw(v, dpp_f<0x141, 0xf>(v));
  if (G >= 16) v = max_raw(v, dpp_f<0x140, 0xf>(v));
  float a, b;
  if (G >= 32) { swap16(v, a, b); v = max_raw(a, b); }
  if (G >= 64) { swap32(v, a, b); v = max_raw(a, b); }
  return v;
}
__device__ __forceinline__ float wave_sum(float v) { return group_sum(v, 64); }

// Row reductions for either geometry.  G <= 64: butterfly inside the lane group.  G == kBlock:
// the whole block owns the row (wave butterfly, then LDS across the four waves); every thread of
// the block must reach the call (uniform control flow).  All lanes get the result.
__device__ __forceinline__ float row_sum(float v, int G) {
  if (G <= 64) return group_sum(v, G);
  __shared__ float lds_s[kBlock / 64];
  v = group_sum(v, 64);
  lds_barrier();  // the previous reduction's readers are done
  if ((threadIdx.x & 63) == 0) lds_s[threadIdx.x >> 6] = v;
  lds_barrier();
  float s = 0.f;
#pragma unroll
  for (int w = 0; w < kBlock / 64; ++w) s += lds_s[w];
  return s;
}
__device__ __forceinline__ float row_max(float v, int G) {
  if (G <= 64) return group_max(v, G);
  __shared__ float lds_m[kBlock / 64];
  v = group_max(v, 64);
  lds_barrier();
  if ((threadIdx.x & 63) == 0) lds_m[threadIdx.x >> 6] = v;
  lds_barrier();
  float s = lds_m[0];
#pragma unroll
  for (int w = 1; w < kBlock / 64; ++w) s = fmaxf(s, lds_m[w]);
  return s;
}

// Wave totals by DPP (v_mov_b32_dpp: no LDS round trip, ~10 cycles a step where a ds_bpermute butterfly step costs
// ~120): xor 1, xor 2 inside quads, half-mirror and mirror inside the 16-lane rows, then row_bcast15 / row_bcast31
// carry the row totals up; lane 63 holds the sum of all 64 lanes in a fixed order and every lane gets it back through
// a readlane.  All 64 lanes must be active.  Used where a latency chain ends in these sums (the closing blocks).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_d(double v) {
  const uint64_t b = __builtin_bit_cast(uint64_t, v);
  const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)b, CTRL, ROW_MASK, 0xf, false);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(b >> 32), CTRL, ROW_MASK, 0xf, false);
  return __builtin_bit_cast(double, ((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ float wave_total(float v) {
  v += dpp_f<0xB1, 0xf>(v);    // quad_perm [1,0,3,2]
  v += dpp_f<0x4E, 0xf>(v);    // quad_perm [2,3,0,1]
  v += dpp_f<0x141, 0xf>(v);   // row_half_mirror
  v += dpp_f<0x140, 0xf>(v);   // row_mirror
  v += dpp_f<0x142, 0xa>(v);   // row_bcast15 into rows 1 and 3
  v += dpp_f<0x143, 0xc>(v);   // row_bcast31 into rows 2 and 3
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ double wave_sum_d(double v) {
  v += dpp_d<0xB1, 0xf>(v);
  v += dpp_d<0x4E, 0xf>(v);
  v += dpp_d<0x141, 0xf>(v);
  v += dpp_d<0x140, 0xf>(v);
  v += dpp_d<0x142, 0xa>(v);
  v += dpp_d<0x143, 0xc>(v);
  const uint64_t b = __builtin_bit_cast(uint64_t, v);
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)b, 63);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(b >> 32), 63);
  return __builtin_bit_cast(double, ((uint64_t)hi << 32) | lo);
}

// ---------------------------------------------------------------------------------------------
// Batch scalars (the *_mean entries): grid-wide sum of N <= 8 per-thread accumulators, finished
// inside the producing kernel (no second launch), deterministic for a given grid.
//   step 0: lane butterfly + LDS across the 4 waves -> bpart[block][k]
//   step 1: the LAST block to arrive (one ticket counter) sums the <= kGridReduce block partials
//           in block order (double), scales, writes out[k].
// Which block arrives last varies from run to run; what it adds, and in which order, does not.
// Hand-off = cdna_hip_programming.md Guideline 16, the all-sc1 form: every shared word (partials,
// counter) is written with a relaxed agent-scope atomic store (write-through, no release fence --
// a per-block agent release fence measured 7x slower end to end), the storing wave drains
// (s_waitcnt vmcnt(0)) before ITS lane 0 takes a ticket with a relaxed agent-scope fetch_add, and
// the closing wave -- one per launch -- does one agent-scope acquire and then plain loads.
// Only wave 0 of a block takes part after the LDS step; no spinning.
// The counter is back at 0 when the kernel ends, so the workspace only has to be zeroed once.
// Tickets: one counter for every block costs ~13 ns per block (same-address atomics serialise:
// 13 us tail at 1024 blocks, measured), so blocks draw from kTicketFan sub-counters on separate
// 64-byte lines and only the last block of each sub-counter draws from the master counter.
// Workspace layout (floats): [0,16) master counter | [16*(1+s), ...) sub-counter s (one per 64 B)
// | [kWsNanFlag] the sticky saw-a-NaN word | [1024, 1024 + kReduceGridMax*8) bpart.
// ---------------------------------------------------------------------------------------------
constexpr int kReduceGridMax = 2048;
constexpr int kTicketFan = 32;
constexpr size_t kWorkspaceFloats = 1024 + (size_t)kReduceGridMax * 8;
// Behind the counters and block partials a caller's workspace holds kRowPartFloats more floats: the per-chunk row
// sums of the one-pass NLL on long rows (csrc/elbo.hip: a row cut into chunks of one block each).  The library's
// own static scratch (vae_nll_elbo_fused_static) has no such area and keeps the block-per-row loop.
constexpr size_t kRowPartFloats = (size_t)1 << 20;
constexpr size_t kWorkspaceFloatsFull = kWorkspaceFloats + kRowPartFloats;

__device__ __forceinline__ void shared_store(float* p, float v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float shared_load(const float* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// "Saw a NaN" word of a workspace (float index kWsNanFlag, read as uint32): the closing wave of every kernel that
// finishes batch means ORs a bit in when a mean comes out NaN -- a NaN in any sample's nll / kld / latent
// parameter makes its batch sum NaN -- so the host can look at one word at the end of an epoch instead of
// synchronising on every step as the reference's nan_check_and_break does (abstract_vae.py:366-369,
// isotropic_gaussian.py:89-90, vrnn.py:427-428).  Sticky until the caller clears it.  `nanbits`: 4 bits per
// output slot, the VAE_NAN_* bit(s) that slot raises.
constexpr int kWsNanFlag = 600;
constexpr uint32_t kNanElbo6 = 0x882111u;   // loss, elbo, nll -> VAE_NAN_NLL | kld -> VAE_NAN_KLD | add, mi -> VAE_NAN_OTHER
constexpr uint32_t kNanGauss2 = 0x44u;      // the two latent-parameter means -> VAE_NAN_LATENT
__device__ __forceinline__ void note_nan(float* __restrict__ ws, float value, uint32_t nanbits, int slot) {
  const uint32_t bits = (nanbits >> (4 * slot)) & 0xfu;
  if (bits && value != value)
    __hip_atomic_fetch_or(reinterpret_cast<unsigned int*>(ws + kWsNanFlag), bits, __ATOMIC_RELAXED,
                          __HIP_MEMORY_SCOPE_AGENT);
}

template <int N>
__device__ __forceinline__ void grid_reduce(float (&acc)[N], float* __restrict__ ws, float scale,
                                            float* __restrict__ out, uint32_t nanbits = 0) {
  static_assert(N <= 8, "at most 8 batch scalars per kernel");
  __shared__ float lds[kBlock / 64][8];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned int* master = reinterpret_cast<unsigned int*>(ws);
  float* bpart = ws + 1024;
  const int grid = gridDim.x;

#pragma unroll
  for (int k = 0; k < N; ++k) {
    const float s = wave_total(acc[k]);
    if (lane == 0) lds[wave][k] = s;
  }
  lds_barrier();
  if (wave != 0) return;
  if (lane < N) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < kBlock / 64; ++w) s += lds[w][lane];
    if (grid == 1) {
      out[lane] = s * scale;
      note_nan(ws, s, nanbits, lane);
    } else {
      shared_store(&bpart[(size_t)blockIdx.x * 8 + lane], s);
    }
  }
  if (grid == 1) return;
  // drain this wave's partial stores, then lane 0 draws a ticket from its sub-counter; the last
  // block of a sub-counter draws from the master; the last there closes the grid
  const int fan = grid < kTicketFan ? grid : kTicketFan;
  const int sub = blockIdx.x % fan;
  const unsigned int sub_count = (unsigned int)(grid / fan + (sub < grid % fan ? 1 : 0));
  unsigned int* subc = master + 16 * (1 + sub);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  unsigned int t = 0;
  if (lane == 0) {
    t = __hip_atomic_fetch_add(subc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t == sub_count - 1) {
      __hip_atomic_store(subc, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      t = __hip_atomic_fetch_add(master, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      t = (t == (unsigned int)(fan - 1)) ? 1u : 0u;
    } else {
      t = 0u;
    }
  }
  t = __builtin_amdgcn_readfirstlane(t);
  asm volatile("" ::: "memory");
  if (!t) return;
  unsigned int* counter = master;

  // this block closes the grid: ONE agent-scope acquire (drops this CU's stale lines; executed by
  // this single block only), then plain 16-byte loads that the compiler can batch -- per-word sc1
  // loads here cost ~1 us each in sequence (16 us tail at 1024 blocks, measured).
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  typedef float f4 __attribute__((ext_vector_type(4)));
  const f4* bp4 = reinterpret_cast<const f4*>(bpart);
  double v[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) v[k] = 0.0;
  // lane j adds blocks j, j+64, ... in order; then a fixed butterfly over lanes
#pragma unroll 4
  for (int b = lane; b < grid; b += 64) {
    const f4 lo = bp4[(size_t)b * 2];
    f4 hi = {0.f, 0.f, 0.f, 0.f};
    if (N > 4) hi = bp4[(size_t)b * 2 + 1];
#pragma unroll
    for (int k = 0; k < N; ++k) v[k] += (double)(k < 4 ? lo[k] : hi[k - 4]);
  }
#pragma unroll
  for (int k = 0; k < N; ++k) {
    const double x = wave_sum_d(v[k]);
    if (lane == 0) {
      out[k] = (float)(x * (double)scale);
      note_nan(ws, (float)x, nanbits, k);
    }
  }
  if (lane == 0) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}


// Thread-level double sums -> out[k] = scale * total, for a closing launch of a few blocks (gridDim.x <= 64: one
// counter): block sums through LDS, one 8-double partial per block in the workspace's partial area, added in block
// order by the block that arrives last.  Same hand-off as grid_reduce.  Every thread of the block calls it.
template <int N>
__device__ __forceinline__ void close_block_sums_d(double (&v)[N], float* __restrict__ ws, float scale,
                                                   float* __restrict__ out, uint32_t nanbits = 0) {
  __shared__ double ldsd[kBlock / 64][8];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < N; ++k) {
    const double x = wave_sum_d(v[k]);
    if (lane == 0) ldsd[wave][k] = x;
  }
  lds_barrier();
  if (wave != 0) return;
  double mine = 0.0;
  if (lane < N) {
#pragma unroll
    for (int w = 0; w < kBlock / 64; ++w) mine += ldsd[w][lane];
  }
  const int grid = gridDim.x;
  if (grid == 1) {
    if (lane < N) {
      out[lane] = (float)(mine * (double)scale);
      note_nan(ws, (float)mine, nanbits, lane);
    }
    return;
  }
  unsigned int* master = reinterpret_cast<unsigned int*>(ws);
  double* cpart = reinterpret_cast<double*>(ws + 1024);
  if (lane < N) __hip_atomic_store(&cpart[(size_t)blockIdx.x * 8 + lane], mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  unsigned int t = 0;
  if (lane == 0)
    t = __hip_atomic_fetch_add(master, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned int)(grid - 1);
  t = __builtin_amdgcn_readfirstlane(t);
  asm volatile("" ::: "memory");
  if (!t) return;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int k = 0; k < N; ++k) {
    double x = lane < grid ? cpart[(size_t)lane * 8 + k] : 0.0;
    x = wave_sum_d(x);
    if (lane == 0) {
      out[k] = (float)(x * (double)scale);
      note_nan(ws, (float)x, nanbits, k);
    }
  }
  if (lane == 0) __hip_atomic_store(master, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Philox4x32-10 counter RNG (Salmon et al. 2011), used when the caller passes no
// noise tensor.  Counter = global element index, key = seed, so a sample's noise
// does not depend on how the batch is sharded over ranks.
struct Philox {
  static constexpr uint32_t kM0 = 0xD2511F53u, kM1 = 0xCD9E8D57u;
  static constexpr uint32_t kW0 = 0x9E3779B9u, kW1 = 0xBB67AE85u;
  __device__ __forceinline__ static void gen(uint64_t seed, uint64_t ctr_lo, uint64_t ctr_hi, uint32_t (&out)[4]) {
    uint32_t c0 = (uint32_t)ctr_lo, c1 = (uint32_t)(ctr_lo >> 32);
    uint32_t c2 = (uint32_t)ctr_hi, c3 = (uint32_t)(ctr_hi >> 32);
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#ifndef VAE_PHILOX_ROUNDS
#define VAE_PHILOX_ROUNDS 10
#endif
#pragma unroll
    for (int r = 0; r < VAE_PHILOX_ROUNDS; ++r) {
      // one 32x32->64 product each (v_mad_u64_u32) instead of a mul_hi + mul_lo pair
      const uint64_t p0 = (uint64_t)kM0 * (uint64_t)c0, p1 = (uint64_t)kM1 * (uint64_t)c2;
      const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
      const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
      uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
      c0 = n0; c1 = n1; c2 = n2; c3 = n3;
      k0 += kW0; k1 += kW1;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
  }
};

// uniform in (0,1): 24 random bits, centred so 0 and 1 are never produced
__device__ __forceinline__ float u01(uint32_t x) { return ((float)(x >> 8) + 0.5f) * (1.0f / 16777216.0f); }

// four N(0,1) draws from four words (two Box-Muller pairs)
__device__ __forceinline__ void normal4(const uint32_t (&w)[4], float (&n)[4]) {
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    // r = sqrt(-2 ln u) from bare v_log_f32 / v_sqrt_f32 (u in (0,1): normal range); the angle is fed to
    // v_sin_f32 / v_cos_f32 directly in turns (their native unit), no 2*pi multiply and no range reduction
    const float r = __builtin_amdgcn_sqrtf(-2.0f * kLn2 * log2_raw(u01(w[2 * p])));
    const float t = u01(w[2 * p + 1]);
    n[2 * p] = r * __builtin_amdgcn_cosf(t);
    n[2 * p + 1] = r * __builtin_amdgcn_sinf(t);
  }
}

// V draws for the elements [elem, elem+V) of a [rows, cols] noise field (elem is
// the global flat index of the first element; V is 1, 4 or 8 and elem % V == 0
// for V > 1).  The stream id separates independent fields of one call.
template <int V>
__device__ __forceinline__ void philox_normal(uint64_t seed, uint64_t stream, uint64_t elem, float (&out)[V]) {
  if constexpr (V == 1) {
    uint32_t w[4];
    float n[4];
    Philox::gen(seed, elem >> 2, stream, w);
    normal4(w, n);
    out[0] = n[elem & 3];
  } else {
#pragma unroll
    for (int q = 0; q < V / 4; ++q) {
      uint32_t w[4];
      float n[4];
      Philox::gen(seed, (elem >> 2) + q, stream, w);
      normal4(w, n);
#pragma unroll
      for (int i = 0; i < 4; ++i) out[4 * q + i] = n[i];
    }
  }
}
// the raw words behind philox_uniform (the same counters): out[i] is the word of element elem + i
template <int V>
__device__ __forceinline__ void philox_words(uint64_t seed, uint64_t stream, uint64_t elem, uint32_t (&out)[V]) {
  if constexpr (V == 1) {
    uint32_t w[4];
    Philox::gen(seed, elem >> 2, stream, w);
    out[0] = w[elem & 3];
  } else {
#pragma unroll
    for (int q = 0; q < V / 4; ++q) {
      uint32_t w[4];
      Philox::gen(seed, (elem >> 2) + q, stream, w);
#pragma unroll
      for (int i = 0; i < 4; ++i) out[4 * q + i] = w[i];
    }
  }
}
template <int V>
__device__ __forceinline__ void philox_uniform(uint64_t seed, uint64_t stream, uint64_t elem, float (&out)[V]) {
  if constexpr (V == 1) {
    uint32_t w[4];
    Philox::gen(seed, elem >> 2, stream, w);
    out[0] = u01(w[elem & 3]);
  } else {
#pragma unroll
    for (int q = 0; q < V / 4; ++q) {
      uint32_t w[4];
      Philox::gen(seed, (elem >> 2) + q, stream, w);
#pragma unroll
      for (int i = 0; i < 4; ++i) out[4 * q + i] = u01(w[i]);
    }
  }
}

// A Philox call offset is a host value, or -- with bit 63 set -- the address of a uint64 device word that holds it
// (VAE_OFFSET_ON_DEVICE, include/vae_amd.h: a HIP graph would freeze a host value into every replay; the word is
// written by vae_rng_next, itself a node of the graph).  Read past the scalar cache (sc1 vector load).
__device__ __forceinline__ uint64_t resolve_offset(uint64_t offset) {
  if (offset >> 63)
    return __hip_atomic_load(reinterpret_cast<const uint64_t*>(offset & ~(1ull << 63)), __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_AGENT);
  return offset;
}

// ---- host-side launch geometry -------------------------------------------------
// Tuning overrides, read from the environment ONCE (first launch) and settable in-process through
// vae_tuning_set (scripts/ab.py); 0 = the built-in choice.
struct Knobs {
  int G = 0;            // VAE_AMD_G        lanes per row
  int U = 0;            // VAE_AMD_U        row-chunks in flight per lane
  int64_t maxgrid = 0;  // VAE_AMD_MAXGRID  grid cap
  int reducegrid = 0;   // VAE_AMD_REDUCEGRID  grid of the kernels that close batch scalars with grid_reduce
};
inline Knobs& knobs() {
  static Knobs k = [] {
    Knobs r;
    if (const char* e = getenv("VAE_AMD_G")) r.G = atoi(e);
    if (const char* e = getenv("VAE_AMD_U")) r.U = atoi(e);
    if (const char* e = getenv("VAE_AMD_MAXGRID")) r.maxgrid = atoll(e);
    if (const char* e = getenv("VAE_AMD_REDUCEGRID")) r.reducegrid = atoi(e);
    return r;
  }();
  return k;
}

struct RowGeom {
  int G;               // lanes per row
  int U;               // row-chunks loaded up front per lane (1, 2 or 4)
  int rows_per_block;  // kBlock / G
  int64_t tiles;       // ceil(rows / rows_per_block)
  int grid;            // min(tiles, max_grid); tiles beyond are grid-strided
};

// Lanes per row: a whole wave (G = 64: 1 KiB contiguous per load/store instruction) once the
// row has more than 64 vectors, else the smallest power of two that leaves every lane two
// vectors per tensor (bf16 L=512, 64 vectors: G = 32 runs 40 % faster than G = 64).  Measured
// (scripts/ab.py): for 784-pixel rows G = 64 beats G = 8/16 by 20-45 % on the store-heavy
// backward although 23 % of the lane slots of the last pass idle.
// Grid: one tile per block (measured: +8 % over a 2048-block grid-stride loop); kernels that
// reduce batch scalars with grid_reduce are capped at kGridReduce blocks.
constexpr int kGridReduce = 1024;  // measured: the per-block ticket costs more than it saves beyond ~4 blocks/CU
constexpr int64_t kGridStream = 1 << 20;

inline RowGeom row_geom(int64_t rows, int64_t vecs_per_row, int64_t max_grid = kGridReduce) {
  const Knobs& kn = knobs();
  RowGeom g;
  int G = 1;
  while (G < 64 && (int64_t)G * 2 < vecs_per_row) G <<= 1;  // >= 2 vectors per lane and tensor in flight
  // a tuning knob never widens the group beyond the row: the kernels assume every lane owns at least one vector
  if (kn.G >= 1 && kn.G <= 64 && (kn.G & (kn.G - 1)) == 0 && kn.G <= vecs_per_row) G = kn.G;
  g.G = G;
  const int64_t iters = (vecs_per_row + G - 1) / G;
  g.U = iters >= 3 ? 4 : (iters == 2 ? 2 : 1);
  if (kn.U == 1 || kn.U == 2 || kn.U == 4) g.U = kn.U;
  if (kn.maxgrid >= 1)
    max_grid = (max_grid <= kReduceGridMax && kn.maxgrid > kReduceGridMax) ? kReduceGridMax : kn.maxgrid;
  if (max_grid == kGridReduce && kn.reducegrid >= 1 && kn.reducegrid <= kReduceGridMax) max_grid = kn.reducegrid;
  g.rows_per_block = kBlock / G;
  g.tiles = (rows + g.rows_per_block - 1) / g.rows_per_block;
  g.grid = (int)(g.tiles < max_grid ? g.tiles : max_grid);
  if (g.grid < 1) g.grid = 1;
  return g;
}

// Another lane group / unroll for a geometry row_geom produced: the per-family rules found by the cold-cache sweeps
// (scripts/sweep_geom.py, profiles/r02_geom_sweep.txt).  Leaves `g` alone while a tuning knob fixes G or U.
inline void regroup(RowGeom& g, int64_t rows, int G, int U, int64_t max_grid) {
  if (knobs().G || knobs().U) return;
  if (g.grid < g.tiles) max_grid = g.grid;   // a cap row_geom took from the tuning knobs
  g.G = G;
  g.U = U;
  g.rows_per_block = kBlock / G;
  g.tiles = (rows + g.rows_per_block - 1) / g.rows_per_block;
  g.grid = (int)(g.tiles < max_grid ? g.tiles : max_grid);
  if (g.grid < 1) g.grid = 1;
}

// Geometry for kernels that keep a whole row in registers (softmax-class: several reductions over
// the same row): the smallest lane group G (<= 64, else the whole block) with ceil(vecs/G) <= 4.
// Returns false when the row does not fit (vecs > 4 * kBlock).
inline bool row_geom_resident(int64_t rows, int64_t vecs_per_row, int64_t max_grid, RowGeom* out) {
  if (vecs_per_row > 4 * kBlock) return false;
  RowGeom g;
  int G = 1;
  while (G < 64 && (int64_t)G * 4 < vecs_per_row) G <<= 1;
  if ((int64_t)G * 4 < vecs_per_row) G = kBlock;
  // tuning knob: a wider group (fewer chunks per lane, fewer registers), never one that no longer holds the row
  const Knobs& kn = knobs();
  if (kn.G > G && kn.G <= 64 && (kn.G & (kn.G - 1)) == 0 && kn.G <= vecs_per_row) G = kn.G;
  g.G = G;
  const int64_t iters = (vecs_per_row + G - 1) / G;
  g.U = iters >= 3 ? 4 : (iters == 2 ? 2 : 1);
  g.rows_per_block = kBlock / G;
  g.tiles = (rows + g.rows_per_block - 1) / g.rows_per_block;
  g.grid = (int)(g.tiles < max_grid ? g.tiles : max_grid);
  if (g.grid < 1) g.grid = 1;
  *out = g;
  return true;
}

// Instantiate `...` with a constexpr U matching the runtime geometry.
#define VAE_SWITCH_U(u_, ...)                   \
  switch (u_) {                                 \
    case 1: { constexpr int U = 1; __VA_ARGS__; } break; \
    case 2: { constexpr int U = 2; __VA_ARGS__; } break; \
    default: { constexpr int U = 4; __VA_ARGS__; } break; \
  }

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace vae
