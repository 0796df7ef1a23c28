// lb_msplit_dev.h - device helpers of the M-split kernels (lb_msplit.hip): fp16 hi/lo staging of a
// tile's activations as MFMA B fragments in LDS, register-resident weight fragments, the cross-wave LayerNorm
// combine, DPP / permlane reductions, the exhaustive range guard, the XCD-aware unit walk.
#pragma once
#include "lb_f16x2.h"
#include "lb_msplit.h"

// ------------------------------------------------------------------------------------------ device helpers
typedef float f32x2m __attribute__((ext_vector_type(2)));
typedef _Float16 h2m __attribute__((ext_vector_type(2)));
typedef unsigned u32x2m __attribute__((ext_vector_type(2)));

// Built with -DLB_MS_STAMPS (debug builds only; every launch then prints them): wave 0 of workgroup 0 stamps the shader
// clock at phase boundaries.  Not in the product library: the run-time test alone cost ~60 scalar instructions and nine
// exec-mask round trips per tile.
#ifdef LB_MS_STAMPS
#define MS_STAMP(k)                                                              \
  do {                                                                           \
    if (a.dbg && blockIdx.x == 0 && threadIdx.x == 0) a.dbg[k] = clock64();      \
  } while (0)
#else
#define MS_STAMP(k) \
  do {              \
  } while (0)
#endif
#define MS_WAVES 4
#define MS_THREADS (MS_WAVES * 64)

// sum / max over the four lanes {n, n+16, n+32, n+48} (same row n, the four k-groups): gfx950 row swaps.
// v_permlane16_swap exchanges the odd 16-lane rows of its first operand with the even rows of the second,
// v_permlane32_swap the upper half of the first with the lower half of the second (tools/permlane_check.hip), so
// with both operands = x the two results are the two addends of the xor-16 / xor-32 butterfly.  Inline asm: through
// __builtin_amdgcn_permlane16_swap hipcc 7.2 emits the swap and then adds the FIRST result to itself (the second
// tied output is lost in a larger kernel); the leading s_nop is the VALU-write -> permlane-read wait the hazard
// recogniser cannot insert inside an asm block.
__device__ __forceinline__ void ms_swap16(float& a, float& b) {
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void ms_swap32(float& a, float& b) {
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ float ms_sum_g(float x) {
  float a = x, b = x;
  ms_swap16(a, b);
  a = b = a + b;
  ms_swap32(a, b);
  return a + b;
}
__device__ __forceinline__ float ms_max_g(float x) {
  float a = x, b = x;
  ms_swap16(a, b);
  a = b = fmaxf(a, b);
  ms_swap32(a, b);
  return fmaxf(a, b);
}
// OR over the four lanes of a row (the range guard's per-row test)
__device__ __forceinline__ uint32_t ms_or_g(uint32_t x) {
  float a = __builtin_bit_cast(float, x), b = a;
  ms_swap16(a, b);
  const uint32_t y = __builtin_bit_cast(uint32_t, a) | __builtin_bit_cast(uint32_t, b);
  a = b = __builtin_bit_cast(float, y);
  ms_swap32(a, b);
  return __builtin_bit_cast(uint32_t, a) | __builtin_bit_cast(uint32_t, b);
}
// max over the whole wave (every lane gets it): rotations inside the 16-lane DPP rows, then the row swaps
__device__ __forceinline__ float ms_wave_max(float m) {
#define MS_ROR(k) m = fmaxf(m, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, m), 0x120 + (k), 0xF, 0xF, false)))
  MS_ROR(8);
  MS_ROR(4);
  MS_ROR(2);
  MS_ROR(1);
#undef MS_ROR
  return ms_max_g(m);
}
__device__ __forceinline__ float ms_absmax4(const f32x4& x) {
  return fmaxf(fmaxf(fabsf(x[0]), fabsf(x[1])), fmaxf(fabsf(x[2]), fabsf(x[3])));
}

// write this wave's 32 features x 16 rows (C layout) as k-block kb of a tile's B-fragment image [kb][part][lane].
// Round-3b instruction diet (the M-split kernels are bound by VALU issue: every wave repeats the per-tile work):
// ReLU as an integer max, the split on the mixed-precision fma (lb_split8v: 12 VALU per 8 values instead of 30),
// and the range guard fed from what the split produces anyway - `orv` receives the OR of the fp16 `hi` bit patterns
// (TINY test, lb_tile_tiny's encoding), `big` the running max |operand| through v_max3 with |abs| modifiers.
template <bool RELU>
__device__ __forceinline__ void ms_stage(f32x4* img, int kb, int lane, const f32x4& x0, const f32x4& x1, uint32_t& orv,
                                         float& big) {
  f32x4 r0 = x0, r1 = x1;
  if (RELU) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float f0 = x0[j], f1 = x1[j];
      r0[j] = __builtin_bit_cast(float, max(__builtin_bit_cast(int, f0), 0));
      r1[j] = __builtin_bit_cast(float, max(__builtin_bit_cast(int, f1), 0));
    }
  }
  h8 hi, lo;
  lb_split8v(r0, r1, hi, lo);
  img[(kb * 2 + 0) * 64 + lane] = __builtin_bit_cast(f32x4, hi);
  img[(kb * 2 + 1) * 64 + lane] = __builtin_bit_cast(f32x4, lo);
  typedef uint32_t u32x4s __attribute__((ext_vector_type(4)));
  const u32x4s u = __builtin_bit_cast(u32x4s, hi);
  orv = (u[0] | u[1]) | (u[2] | u[3]);
  asm("v_max3_f32 %0, |%1|, |%2|, %0\n\t"
      "v_max3_f32 %0, |%3|, |%4|, %0\n\t"
      "v_max3_f32 %0, |%5|, |%6|, %0\n\t"
      "v_max3_f32 %0, |%7|, |%8|, %0"
      : "+v"(big)
      : "v"(r0[0]), "v"(r0[1]), "v"(r0[2]), "v"(r0[3]), "v"(r1[0]), "v"(r1[1]), "v"(r1[2]), "v"(r1[3]));
}

// acc[c] += W_c^T B over NKB k-blocks for this wave's NB output blocks: weights from registers, B fragments from the
// tile's LDS image; the NB accumulate chains are interleaved pass by pass (lo*hi, hi*lo, hi*hi)
template <int NKB, int NB>
__device__ __forceinline__ void ms_gemm(const f32x4* img, int lane, const h8 (&wh)[NB][NKB], const h8 (&wl)[NB][NKB],
                                        f32x4 (&acc)[NB]) {
#pragma unroll
  for (int kb = 0; kb < NKB; ++kb) {
    const h8 bh = __builtin_bit_cast(h8, img[(kb * 2 + 0) * 64 + lane]);
    const h8 bl = __builtin_bit_cast(h8, img[(kb * 2 + 1) * 64 + lane]);
#pragma unroll
    for (int c = 0; c < NB; ++c) acc[c] = MFMA16H(wl[c][kb], bh, acc[c]);
#pragma unroll
    for (int c = 0; c < NB; ++c) acc[c] = MFMA16H(wh[c][kb], bl, acc[c]);
#pragma unroll
    for (int c = 0; c < NB; ++c) acc[c] = MFMA16H(wh[c][kb], bh, acc[c]);
  }
}
template <int NKB, int NB>
__device__ __forceinline__ void ms_wload(const f32x4* wb, int mb0, h8 (&wh)[NB][NKB], h8 (&wl)[NB][NKB]) {
#pragma unroll
  for (int c = 0; c < NB; ++c)
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
      wh[c][kb] = __builtin_bit_cast(h8, wb[(((mb0 + c) * NKB + kb) * 2 + 0) * 64]);
      wl[c][kb] = __builtin_bit_cast(h8, wb[(((mb0 + c) * NKB + kb) * 2 + 1) * 64]);
    }
}

// LayerNorm over the 128 features of a row that are spread over the 4 waves (32 each): every wave contributes
// (sum, M2 about its own mean) per row; combined with the parallel-variance update (Chan et al.).
__device__ __forceinline__ f32x2m ms_ln_local(const f32x4& x0, const f32x4& x1) {
  const float s = ms_sum_g(((x0[0] + x0[1]) + (x0[2] + x0[3])) + ((x1[0] + x1[1]) + (x1[2] + x1[3])));
  const float mu = s * 0.03125f;
  float m2 = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float d0 = x0[j] - mu, d1 = x1[j] - mu;
    m2 += d0 * d0 + d1 * d1;
  }
  return f32x2m{s, ms_sum_g(m2)};
}
// red: [16 rows][4 waves] (sum, M2) of one tile
__device__ __forceinline__ void ms_ln_combine(const f32x2m* red, int n, float inv_d, float pad, float& mean,
                                              float& rs) {
  const f32x4* r4 = reinterpret_cast<const f32x4*>(red + n * 4);
  const f32x4 a = r4[0], b = r4[1];
  const float sw[4] = {a[0], a[2], b[0], b[2]}, mw[4] = {a[1], a[3], b[1], b[3]};
  mean = ((sw[0] + sw[1]) + (sw[2] + sw[3])) * inv_d;
  float m2 = 0.f;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float dm = sw[q] * 0.03125f - mean;
    m2 += mw[q] + 32.f * (dm * dm);
  }
  // a latent narrower than 128 is zero-padded: each padded entry contributed mean^2 to the centred sum
  rs = 1.0f / sqrtf(fmaxf(m2 - pad * (mean * mean), 0.f) * inv_d + 1e-5f);
}

// running range-guard state of a wave (see the header).  LARGE: `big` = max |x| over every GEMM operand element this
// wave staged (ms_stage; a value that leaves the fp16 range is seen BEFORE its `hi` becomes inf).  TINY, round 4: per
// ROW of every tile and operand (lb_f16x2.h: lb_rows_tiny has the why) - the OR of the `hi` bit patterns of a row says
// whether all of its elements are below 2^-11 (exponent bits 14:12 clear) and not all zero; the 4 waves hold 32 features
// each of the tile's 16 rows, so each posts its 16 per-row ORs (`code`: lanes 0-15 write them) and after the barrier
// lane n combines the four words of row n (`tile_codes`).
#define MS_GUARD_WORDS 16  // ints per (parity, operand, tile) slot: one OR word per row
struct ms_guard {
  float big;
  int flags;
  // every lane ORs its word into the slot of its row (LDS atomic: 4 lanes per address; the four waves add theirs) - no
  // cross-lane reduction, no ballot in the latency chain of the B = 1 launches (a permlane-based version of this test
  // cost TGV2D-2.5k B = 1 16 %: 0.26 -> 0.30 ms per step)
  __device__ __forceinline__ void post(int* slot, uint32_t orv) const { atomicOr(&slot[threadIdx.x & 15], (int)orv); }
  // after the barrier: lane n tests row n; the slot of the OTHER tile parity (last read one iteration ago, next written
  // one iteration ahead, barriers in between) is cleared for its next use
  __device__ __forceinline__ void tile_codes(const int* slot, int* slot_other) {
    const uint32_t r = (uint32_t)slot[threadIdx.x & 15];
    if ((r & 0x70007000u) == 0u && (r & 0x7fff7fffu) != 0u) flags |= LB_MATH_TINY;
    if (threadIdx.x < 16) slot_other[threadIdx.x] = 0;
  }
  __device__ __forceinline__ void commit(const lb_ctrl* ctrl, int lane) {
    const float m = ms_wave_max(big);
    int f = flags;
    if (!(m < 32768.f)) f |= LB_MATH_LARGE;  // close to the fp16 range, inf
    const bool any_large = __any(f & LB_MATH_LARGE), any_tiny = __any(f & LB_MATH_TINY);
    f = (any_large ? LB_MATH_LARGE : 0) | (any_tiny ? LB_MATH_TINY : 0);
    if (lane == 0 && f) lb_raise_math(ctrl, f);
  }
};

// XCD-aware unit walk (block b runs on XCD b % 8): every XCD owns a contiguous eighth of the units
struct ms_walk {
  int q, stride, n_iter, q_last;
  __device__ __forceinline__ bool init(int nq) {
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    stride = gridDim.x >> 3;
    const int q_lo = (int)(((int64_t)nq * xcd) >> 3), q_hi = (int)(((int64_t)nq * (xcd + 1)) >> 3);
    q = q_lo + slot;
    if (q >= q_hi) return false;
    n_iter = (q_hi - 1 - q) / stride + 1;
    q_last = q + (n_iter - 1) * stride;
    return true;
  }
};

