// lb_f16x2.h - device helpers shared by the f16x2 network kernels of round 2 (lb_edge16v.hip,
// lb_node16s.hip): fp16 hi/lo split on the mixed-precision fma, the phase-pipelined MFMA block loop
// over LDS-resident packed weights, packed-fp32 LayerNorm.  See lb_edge16v.hip for the why.
#pragma once
#include "lb_device.h"

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) const f32x4* lds_cptr;
#define MFMA16H(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16((a), (b), (c), 0, 0, 0)
#define SB() __builtin_amdgcn_sched_barrier(0)
// s_waitcnt immediate (gfx9 encoding) that waits only on lgkmcnt <= n
#define LB_WAIT_LGKM(n) (0xC07F | ((n) << 8))
typedef float f32x2v __attribute__((ext_vector_type(2)));
// packed fp32 helpers on the two halves of an f32x4 (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32: one
// issue slot for two lanes-worth of work; the halves of an f32x4 are adjacent register pairs)
__device__ __forceinline__ f32x2v lb_lo2(const f32x4& x) { return f32x2v{x[0], x[1]}; }
__device__ __forceinline__ f32x2v lb_hi2(const f32x4& x) { return f32x2v{x[2], x[3]}; }
__device__ __forceinline__ f32x4 lb_cat2(f32x2v a, f32x2v b) { return f32x4{a[0], a[1], b[0], b[1]}; }
__device__ __forceinline__ f32x4 lb_pk_add(const f32x4& a, const f32x4& b) {
  return lb_cat2(lb_lo2(a) + lb_lo2(b), lb_hi2(a) + lb_hi2(b));
}

// Range guard of the f16x2 arithmetic (sampled: one operand tile per wave and launch).  A value x is
// carried as fp16 hi + fp16 lo: |x| >= 65504 overflows hi, and below 2^-14 * 2^11 the lo half leaves
// the fp16 normal range (absolute floor 2^-25 instead of a relative 2^-22).  LayerNorm keeps GNS
// latents O(1), but a trained checkpoint is not bound to: the kernels raise lb_ctrl::math_flags and
// the host repeats the work on the exact-fp32 MFMA path (lb_api.hip: lb_math_check).
__device__ __forceinline__ void lb_range_probe(const lb_ctrl* ctrl, const f32x4* v, int nvec) {
  float m = 0.f;
  for (int i = 0; i < nvec; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) m = fmaxf(m, fabsf(v[i][j]));
  bool bad = !(m == m) || m >= 32768.f;  // NaN / inf / close to the fp16 range
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
  const bool any_bad = __any(bad);
  if ((threadIdx.x & 63) == 0) {
    int f = 0;
    if (any_bad || m >= 32768.f) f |= LB_MATH_LARGE;
    if (m > 0.f && m < 0.0009765625f) f |= LB_MATH_TINY;
    if (f) lb_raise_math(ctrl, f);
  }
}

// Exhaustive TINY test of the range guard.  `orv` = OR of the fp16 `hi` bit patterns of every B-operand element a
// lane fed into one GEMM of one tile.  Exponent bits 14:12 of a half are clear iff |hi| < 2^-11; if that holds for
// EVERY element of the tile (all lanes) and the operand is not identically zero, the `lo` halves of the whole tile
// are subnormal (absolute 2^-25 floor instead of a relative 2^-22): raise LB_MATH_TINY.  (LARGE needs no exhaustive
// test: an overflowing `hi` is inf, travels through every later layer and is caught by the decoder's check.)
__device__ __forceinline__ bool lb_tile_tiny(uint32_t orv) {
  return !__any((orv & 0x70007000u) != 0u) && __any((orv & 0x7fff7fffu) != 0u);
}

// Per-ROW test (round 4).  A row (edge / node) whose GEMM operand is uniformly small carries
// `lo` halves that are fp16 SUBNORMALS: |x - hi - lo| <= 2^-25 absolute, i.e. 2^-25 / max|row| relative to the row -
// harmless next to a bias, but a following LayerNorm rescales the row to O(1) and the error with it (a row of
// post-ReLU hiddens ~1e-6 with zero bias came out 3 % off).  The tile-wide test above cannot see one such row among
// fifteen ordinary ones.  `orv` = OR of the `hi` patterns of the values THIS lane fed for row n = lane & 15 (its 8
// features of the sampled k-group, or its 32 of the row): exponent bits 14:12 clear on all four lanes of the row <=>
// every sampled |x| < 2^-11; an all-zero sample is exact and does not count.  Threshold: above 2^-11 the row's error is
// <= 2^-14 = 6e-5 of ITS OWN scale, one message among the ~10 a receiver sums - inside the 1e-5 bar on the accelerations;
// a first version flagged rows below 2^-7 (bound 3.8e-6) and fired on every synthetic rollout in which two particles
// pass within 0.5 % of the cutoff of each other (their edge-feature row, hence its encoder hidden row, is that small):
// three fp32 steps per rollout, TGV2D B = 1 0.27 -> 0.30 ms per step.  Sampling one k-group: a row that is tiny in all
// 128 features is tiny in these 32; the converse (a false alarm) costs one step in fp32.
__device__ __forceinline__ bool lb_rows_tiny(uint32_t orv) {
  // OR over the four lanes {n, n+16, n+32, n+48} of row n: gfx950 row swaps (v_permlane16_swap exchanges the odd
  // 16-lane rows of its first operand with the even rows of the second, v_permlane32_swap the upper half of the first
  // with the lower half of the second; inline asm with the wait states the hazard recogniser cannot add - lb_msplit_dev.h)
  uint32_t a = orv, b = orv;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  a = b = a | b;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  const uint32_t r = a | b;
  return __any((r & 0x70007000u) == 0u && (r & 0x7fff7fffu) != 0u);
}

// hi = fp16(x) (RNE), lo = fp16(x - hi) for 8 values: 4 v_cvt_pk_f16_f32 + 8 v_fma_mix{lo,hi}_f16
// (the mixed-precision fma evaluates x*1.0 - float(hi) exactly in fp32 and rounds once to fp16).
// One asm block: the hazard recogniser does not look inside inline asm, so the block ends with the
// two wait states a VALU result needs before an MFMA may read it as SrcA/B.
__device__ __forceinline__ void lb_split8v(const f32x4& x0, const f32x4& x1, h8& hi, h8& lo) {
  union {
    h8 v;
    uint32_t u[4];
  } H, L;
  asm("v_cvt_pk_f16_f32 %0, %8, %9\n"
      "v_cvt_pk_f16_f32 %1, %10, %11\n"
      "v_cvt_pk_f16_f32 %2, %12, %13\n"
      "v_cvt_pk_f16_f32 %3, %14, %15\n"
      "v_fma_mixlo_f16 %4, %8, 1.0, -%0 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n"
      "v_fma_mixlo_f16 %5, %10, 1.0, -%1 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n"
      "v_fma_mixlo_f16 %6, %12, 1.0, -%2 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n"
      "v_fma_mixlo_f16 %7, %14, 1.0, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n"
      "v_fma_mixhi_f16 %4, %9, 1.0, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n"
      "v_fma_mixhi_f16 %5, %11, 1.0, -%1 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n"
      "v_fma_mixhi_f16 %6, %13, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n"
      "v_fma_mixhi_f16 %7, %15, 1.0, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n"
      "s_nop 1"
      : "=&v"(H.u[0]), "=&v"(H.u[1]), "=&v"(H.u[2]), "=&v"(H.u[3]), "=&v"(L.u[0]), "=&v"(L.u[1]), "=&v"(L.u[2]),
        "=&v"(L.u[3])
      : "v"(x0[0]), "v"(x0[1]), "v"(x0[2]), "v"(x0[3]), "v"(x1[0]), "v"(x1[1]), "v"(x1[2]), "v"(x1[3]));
  hi = H.v;
  lo = L.v;
}

// acc[0..7] += W^T * v over NP k-steps of 32 (f16x2: lo*hi + hi*lo + hi*hi), phase-pipelined LDS reads.
// wbase: this lane's LDS pointer to fragment (p 0, mbo 0, part 0); fragment (p, mbo, part) sits
// ((p*8 + mbo)*2 + part)*64 f32x4 further.  RELU applies max(x, 0) to v while it is split.
// GUARD: the fp16 `hi` halves of the B operand are OR-ed into `orv` while they are split (two v_or3 per k-step): the
// exhaustive TINY test of the range guard (lb_tile_tiny below).
// GUARD: 0 none; 1 the FIRST k-group only (32 of the features of every row: the per-row test lb_rows_tiny, round 4's
// default - two v_or3 per GEMM); 2 every k-group (lb_tile_tiny / lb_rows_tiny on all features, LB_GUARD=full).
template <bool RELU, int NP = 4, int GUARD = 0>
__device__ __forceinline__ void lb_gemm16v_r03(lds_cptr wbase, const f32x4 (&v)[2 * NP], f32x4 (&acc)[8], uint32_t* orv = nullptr) {
  auto note = [&](const h8& h, bool first = false) {
    if constexpr (GUARD == 2 || GUARD == 1) {
      if (GUARD == 1 && !first) return;
      typedef uint32_t u32x4g __attribute__((ext_vector_type(4)));
      const u32x4g u = __builtin_bit_cast(u32x4g, h);
      *orv |= (u[0] | u[1]) | (u[2] | u[3]);
    }
  };
  auto frag = [&](int p, int mbo, int part) -> h8 {
    return __builtin_bit_cast(h8, wbase[((p * 8 + mbo) * 2 + part) * 64]);
  };
  auto relu4 = [&](const f32x4& x) -> f32x4 {
    if (!RELU) return x;
    f32x4 r;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float f = x[j];  // (bit_cast of a vector-element lvalue reads element 0: copy first)
      r[j] = __builtin_bit_cast(float, max(__builtin_bit_cast(int, f), 0));
    }
    return r;
  };
  h8 X[4], Y[4];  // X: lo fragments, Y: hi fragments of the current block (4 output blocks)
  h8 bh, bl, nbh, nbl;
#pragma unroll
  for (int c = 0; c < 4; ++c) X[c] = frag(0, c, 1);
#pragma unroll
  for (int c = 0; c < 4; ++c) Y[c] = frag(0, c, 0);
  lb_split8v(relu4(v[0]), relu4(v[1]), bh, bl);
  note(bh, true);
  SB();
#pragma unroll
  for (int blk = 0; blk < 2 * NP; ++blk) {
    const int p = blk >> 1, q = blk & 1;
    const int np = (blk + 1) >> 1, nq = (blk + 1) & 1;
    f32x4* a4 = &acc[4 * q];
    // phase 1: lo * hi.  ONE wait for the four `lo` fragments (the four `hi` reads issued after them may
    // still be in flight): left alone the compiler waits before every MFMA, ~100 s_waitcnt per tile
    // in a kernel that is bound by instruction issue
    __builtin_amdgcn_s_waitcnt(LB_WAIT_LGKM(4));
#pragma unroll
    for (int c = 0; c < 4; ++c) a4[c] = MFMA16H(X[c], bh, a4[c]);
    SB();
    if (blk < 2 * NP - 1) {
#pragma unroll
      for (int c = 0; c < 4; ++c) X[c] = frag(np, 4 * nq + c, 1);
    }
    // the next k-step's operand is split while this block's MFMAs run
    if (q == 1 && p < NP - 1) {
      lb_split8v(relu4(v[2 * p + 2]), relu4(v[2 * p + 3]), nbh, nbl);
      note(nbh);
    }
    // phase 2: hi * lo, hi * hi (wait for the `hi` fragments; the next block's `lo` reads stay in flight)
    if (blk < 2 * NP - 1)
      __builtin_amdgcn_s_waitcnt(LB_WAIT_LGKM(4));
    else
      __builtin_amdgcn_s_waitcnt(LB_WAIT_LGKM(0));
#pragma unroll
    for (int c = 0; c < 4; ++c) a4[c] = MFMA16H(Y[c], bl, a4[c]);
#pragma unroll
    for (int c = 0; c < 4; ++c) a4[c] = MFMA16H(Y[c], bh, a4[c]);
    SB();
    if (blk < 2 * NP - 1) {
#pragma unroll
      for (int c = 0; c < 4; ++c) Y[c] = frag(np, 4 * nq + c, 0);
    }
    if (q == 1 && p < NP - 1) {
      bh = nbh;
      bl = nbl;
    }
    SB();
  }
}


// ---- round 4: the same block loop with the NEXT operand's split issued one instruction at a time in the shadow of the
// MFMAs.  tools/issue_bench (profiles/r04_issue_bench.txt): in ONE wave a 16x16x32 f16 MFMA hides two plain VALU
// instructions or one half-rate one (v_fma_mix*, v_exp, v_rcp) completely (18.3 / 18.0 cycles per slot against 16.5
// bare); the third plain one costs 4 cycles, every further half-rate one 8.  Round 2-3 issued the split as ONE 12-
// instruction block between two MFMA phases: ~85 cycles in series with the matrix pipe, four times per GEMM.  Here the
// 24 MFMAs of k-step p carry, one per slot: [8 integer-max of the ReLU, two per slot] 4 v_cvt_pk_f16_f32, 4
// v_fma_mixlo_f16, 4 v_fma_mixhi_f16 [, 2 v_or3 of the range guard] of k-step p + 1.  Every filler is its own asm
// statement pinned behind its MFMA by a sched_barrier; a split result is first read by an MFMA >= 8 instructions later
// (no VALU -> MFMA wait states needed), mixhi reads the register mixlo wrote 4 slots earlier.
#ifndef LB_GEMM_INTERLEAVE
#define LB_GEMM_INTERLEAVE 1
#endif
struct lb_split_regs {
  uint32_t H[4], L[4];
  f32x4 r0, r1;  // (ReLU'd) operand pair being split
};
template <bool RELU, int GUARD>
__device__ __forceinline__ void lb_split_slot(int s, const f32x4& x0, const f32x4& x1, lb_split_regs& R, uint32_t* orv) {
  // s = 0 .. 23 (compile-time after unrolling)
  if (s < 4) {
    if constexpr (RELU) {  // two integer max per slot
      const float a = s < 2 ? x0[2 * s] : x1[2 * (s - 2)], b = s < 2 ? x0[2 * s + 1] : x1[2 * (s - 2) + 1];
      const float ra = __builtin_bit_cast(float, max(__builtin_bit_cast(int, a), 0));
      const float rb = __builtin_bit_cast(float, max(__builtin_bit_cast(int, b), 0));
      if (s < 2) {
        R.r0[2 * s] = ra;
        R.r0[2 * s + 1] = rb;
      } else {
        R.r1[2 * (s - 2)] = ra;
        R.r1[2 * (s - 2) + 1] = rb;
      }
    }
  } else if (s < 8) {
    const int i = s - 4;
    const f32x4& y = i < 2 ? (RELU ? R.r0 : x0) : (RELU ? R.r1 : x1);
    const float a = y[2 * (i & 1)], b = y[2 * (i & 1) + 1];
    asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(R.H[i]) : "v"(a), "v"(b));
  } else if (s < 12) {
    const int i = s - 8;
    const f32x4& y = i < 2 ? (RELU ? R.r0 : x0) : (RELU ? R.r1 : x1);
    const float a = y[2 * (i & 1)];
    asm volatile("v_fma_mixlo_f16 %0, %1, 1.0, -%2 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=&v"(R.L[i]) : "v"(a), "v"(R.H[i]));
  } else if (s < 16) {
    const int i = s - 12;
    const f32x4& y = i < 2 ? (RELU ? R.r0 : x0) : (RELU ? R.r1 : x1);
    const float b = y[2 * (i & 1) + 1];
    asm volatile("v_fma_mixhi_f16 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(R.L[i]) : "v"(b), "v"(R.H[i]));
  } else if (s == 16) {
    if constexpr (GUARD == 2) *orv |= (R.H[0] | R.H[1]) | (R.H[2] | R.H[3]);
  }
}
template <bool RELU, int NP = 4, int GUARD = 0>
__device__ __forceinline__ void lb_gemm16v(lds_cptr wbase, const f32x4 (&v)[2 * NP], f32x4 (&acc)[8], uint32_t* orv = nullptr) {
  if constexpr (!LB_GEMM_INTERLEAVE || NP == 1) {
    lb_gemm16v_r03<RELU, NP, GUARD>(wbase, v, acc, orv);
    return;
  } else {
  auto frag = [&](int p, int mbo, int part) -> h8 {
    return __builtin_bit_cast(h8, wbase[((p * 8 + mbo) * 2 + part) * 64]);
  };
  h8 X[4], Y[4];  // X: lo fragments, Y: hi fragments of the current block (4 output blocks)
  h8 bh, bl;
  lb_split_regs R;
#pragma unroll
  for (int c = 0; c < 4; ++c) X[c] = frag(0, c, 1);
#pragma unroll
  for (int c = 0; c < 4; ++c) Y[c] = frag(0, c, 0);
  {  // the first operand pair is split up front (nothing to hide it under)
    f32x4 a0 = v[0], a1 = v[1];
    if constexpr (RELU) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float f0 = a0[j], f1 = a1[j];
        a0[j] = __builtin_bit_cast(float, max(__builtin_bit_cast(int, f0), 0));
        a1[j] = __builtin_bit_cast(float, max(__builtin_bit_cast(int, f1), 0));
      }
    }
    lb_split8v(a0, a1, bh, bl);
    if constexpr (GUARD != 0) {
      typedef uint32_t u32x4g __attribute__((ext_vector_type(4)));
      const u32x4g u = __builtin_bit_cast(u32x4g, bh);
      *orv |= (u[0] | u[1]) | (u[2] | u[3]);
    }
  }
  SB();
#pragma unroll
  for (int blk = 0; blk < 2 * NP; ++blk) {
    const int p = blk >> 1, q = blk & 1;
    const int np = (blk + 1) >> 1, nq = (blk + 1) & 1;
    const bool fill = p < NP - 1;       // k-step p + 1 exists: its split rides in this k-step's 24 slots
    const int s0 = 12 * q;
    f32x4* a4 = &acc[4 * q];
    __builtin_amdgcn_s_waitcnt(LB_WAIT_LGKM(4));
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      a4[c] = MFMA16H(X[c], bh, a4[c]);
      if (fill) lb_split_slot<RELU, GUARD>(s0 + c, v[2 * p + 2], v[2 * p + 3], R, orv);
      SB();
    }
    if (blk < 2 * NP - 1) {
#pragma unroll
      for (int c = 0; c < 4; ++c) X[c] = frag(np, 4 * nq + c, 1);
    }
    if (blk < 2 * NP - 1)
      __builtin_amdgcn_s_waitcnt(LB_WAIT_LGKM(4));
    else
      __builtin_amdgcn_s_waitcnt(LB_WAIT_LGKM(0));
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      a4[c] = MFMA16H(Y[c], bl, a4[c]);
      if (fill) lb_split_slot<RELU, GUARD>(s0 + 4 + c, v[2 * p + 2], v[2 * p + 3], R, orv);
      SB();
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      a4[c] = MFMA16H(Y[c], bh, a4[c]);
      if (fill) lb_split_slot<RELU, GUARD>(s0 + 8 + c, v[2 * p + 2], v[2 * p + 3], R, orv);
      SB();
    }
    if (blk < 2 * NP - 1) {
#pragma unroll
      for (int c = 0; c < 4; ++c) Y[c] = frag(np, 4 * nq + c, 0);
    }
    if (q == 1 && fill) {  // the split of k-step p + 1 is complete: it becomes the B operand
      typedef uint32_t u32x4s __attribute__((ext_vector_type(4)));
      bh = __builtin_bit_cast(h8, u32x4s{R.H[0], R.H[1], R.H[2], R.H[3]});
      bl = __builtin_bit_cast(h8, u32x4s{R.L[0], R.L[1], R.L[2], R.L[3]});
    }
    SB();
  }
  }
}

// LayerNorm over the 128 features of every edge (32 registers x the 4 lanes of a DPP-row column),
// hk.LayerNorm(axis=-1, eps 1e-5) of models/utils.py:112, in packed fp32: pre -> y.
// lns / lno: this lane's LDS pointers to the scale / offset vectors (entry 4*mb = features 16mb+4g..).
template <bool ENABLE = true>
__device__ __forceinline__ void lb_layernorm16(f32x4 (&pre)[8], lds_cptr lns, lds_cptr lno, f32x4 (&y)[8],
                                               float inv_d = 1.0f / 128.0f, float pad = 0.f) {
  if constexpr (!ENABLE) {
#pragma unroll
    for (int mb = 0; mb < 8; ++mb) y[mb] = pre[mb];
    return;
  }
  f32x2v s2 = {0.f, 0.f};
#pragma unroll
  for (int mb = 0; mb < 8; ++mb) s2 = s2 + (lb_lo2(pre[mb]) + lb_hi2(pre[mb]));
  float sm = s2[0] + s2[1];
  sm += __shfl_xor(sm, 16);
  sm += __shfl_xor(sm, 32);
  const float mean = sm * inv_d;
  const f32x2v m2 = {mean, mean};
  f32x2v v2 = {0.f, 0.f};
#pragma unroll
  for (int mb = 0; mb < 8; ++mb) {
    const f32x2v dl = lb_lo2(pre[mb]) - m2, dh = lb_hi2(pre[mb]) - m2;
    v2 = __builtin_elementwise_fma(dl, dl, v2);
    v2 = __builtin_elementwise_fma(dh, dh, v2);
    pre[mb] = lb_cat2(dl, dh);
  }
  float vs = v2[0] + v2[1];
  vs += __shfl_xor(vs, 16);
  vs += __shfl_xor(vs, 32);
  // a latent narrower than 128 is zero-padded: the padded entries each contributed mean^2
  const float rs = 1.0f / sqrtf(fmaxf(vs - pad * (mean * mean), 0.f) * inv_d + 1e-5f);
  const f32x2v r2 = {rs, rs};
#pragma unroll
  for (int mb = 0; mb < 8; ++mb) {
    const f32x4 sc = lns[4 * mb], of = lno[4 * mb];
    y[mb] = lb_cat2(__builtin_elementwise_fma(lb_lo2(sc) * r2, lb_lo2(pre[mb]), lb_lo2(of)),
                    __builtin_elementwise_fma(lb_hi2(sc) * r2, lb_hi2(pre[mb]), lb_hi2(of)));
  }
}

// Where does a receiver's segment sum go?  The node kernel (lb_load_agg16) reads `agg[r]` when all
// edges of r lie in one 16-edge tile and otherwise, for every tile the row touches, the partial slot
// `part[tile][k0 <= 16*tile ? 0 : 1]` (k0 = row_ptr[r]).  Both facts are visible from inside the tile:
// slot 0 <=> the segment contains the tile's first lane; complete <=> the edge before the tile (if the
// segment starts at lane 0) and the edge after it (if it ends at lane 15) belong to other receivers.
// lb_edge_probe fetches those two receivers in ONE vector load (even lanes: edge 16t-1, odd lanes:
// edge 16t+16) instead of two row_ptr gathers per lane.
__device__ __forceinline__ int lb_edge_probe(const int32_t* __restrict__ receivers, int t, int lane, int E) {
  int idx = t * 16 - 1 + 17 * (lane & 1);
  idx = idx < 0 ? 0 : (idx < E ? idx : E - 1);
  return receivers[idx];
}
__device__ __forceinline__ bool lb_seg_complete(int rb, int rr, int segstart, int n, int t, int E, int& slot01) {
  const int r_before = __builtin_amdgcn_readlane(rb, 0), r_after = __builtin_amdgcn_readlane(rb, 1);
  const bool starts_before = segstart == 0 && t > 0 && r_before == rr;
  const bool ends_after = n == 15 && t * 16 + 16 < E && r_after == rr;
  slot01 = segstart == 0 ? 0 : 1;
  return !starts_before && !ends_after;
}

// x += row_shr:k(x) * m for 8 registers and k = 1, 2, 4, 8 in a fixed order: one v_fmac_f32_dpp per
// register and step; a register is read through DPP again only 8 instructions after it was written
// (the VALU-write -> DPP-read hazard needs 2 wait states; inline asm is invisible to the hazard
// recogniser, hence the fixed order and the leading s_nop).
__device__ __forceinline__ void lb_scan8(f32x4& a, f32x4& b, float m1, float m2, float m4, float m8) {
  asm volatile(
      "s_nop 1\n"
      "v_fmac_f32_dpp %0, %0, %8 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
      "v_fmac_f32_dpp %1, %1, %8 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
      "v_fmac_f32_dpp %2, %2, %8 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
      "v_fmac_f32_dpp %3, %3, %8 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
      "v_fmac_f32_dpp %4, %4, %8 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
      "v_fmac_f32_dpp %5, %5, %8 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
      "v_fmac_f32_dpp %6, %6, %8 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
      "v_fmac_f32_dpp %7, %7, %8 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
      "v_fmac_f32_dpp %0, %0, %9 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
      "v_fmac_f32_dpp %1, %1, %9 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
      "v_fmac_f32_dpp %2, %2, %9 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
      "v_fmac_f32_dpp %3, %3, %9 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
      "v_fmac_f32_dpp %4, %4, %9 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
      "v_fmac_f32_dpp %5, %5, %9 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
      "v_fmac_f32_dpp %6, %6, %9 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
      "v_fmac_f32_dpp %7, %7, %9 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
      "v_fmac_f32_dpp %0, %0, %10 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
      "v_fmac_f32_dpp %1, %1, %10 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
      "v_fmac_f32_dpp %2, %2, %10 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
      "v_fmac_f32_dpp %3, %3, %10 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
      "v_fmac_f32_dpp %4, %4, %10 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
      "v_fmac_f32_dpp %5, %5, %10 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
      "v_fmac_f32_dpp %6, %6, %10 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
      "v_fmac_f32_dpp %7, %7, %10 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
      "v_fmac_f32_dpp %0, %0, %11 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
      "v_fmac_f32_dpp %1, %1, %11 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
      "v_fmac_f32_dpp %2, %2, %11 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
      "v_fmac_f32_dpp %3, %3, %11 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
      "v_fmac_f32_dpp %4, %4, %11 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
      "v_fmac_f32_dpp %5, %5, %11 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
      "v_fmac_f32_dpp %6, %6, %11 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
      "v_fmac_f32_dpp %7, %7, %11 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
      "s_nop 1\n"
      : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3])
      : "v"(m1), "v"(m2), "v"(m4), "v"(m8));
}
