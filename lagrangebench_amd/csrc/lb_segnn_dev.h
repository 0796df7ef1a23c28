// lb_segnn_dev.h - device helpers shared by the fused SEGNN kernels (lb_segnn_msg.hip: message / update,
// lb_segnn_node.hip: embedding / readout): one lmax-1 tensor-product block on a register-resident operand row in the
// f16x2 MFMA scheme, and the e3nn gate.  Conventions: oracle/segnn_oracle.py (A1-A6); reference:
// lagrangebench/models/segnn.py:44-181.
#pragma once
#include "lb_f16x2.h"

// the library is built with -ffp-contract=off for the fp64 geometry (bit-exact against the oracle); the network
// arithmetic of this file is compared at 1e-5 and wants its multiply-adds fused
#pragma clang fp contract(fast)

static constexpr float SG_Y0 = 0.28209479177387814f;
static constexpr float SG_Y1 = 0.4886025119029199f;
static constexpr float SG_C_SILU = 1.6765620f;
static constexpr float SG_C_SIGMOID = 1.8462292f;
static constexpr float SG_NL2E = -1.4426950408889634f;      // -log2(e): z = SG_NL2E * x
static constexpr float SG_K_SILU = SG_C_SILU / SG_NL2E;     // C_silu * silu(x) = SG_K_SILU * z * sigma(z)

// (tools/sg_msg_bench ablation: NOMFMA drops the matrix instructions of sg_operand)
template <bool NOMFMA>
__device__ __forceinline__ f32x4 sg_mfma(const h8& a, const h8& b, const f32x4& c) {
  if constexpr (NOMFMA) {
    asm volatile("" ::"v"(a), "v"(b));  // operands stay alive: LDS reads and splits are kept
    return c;
  } else {
    return MFMA16H(a, b, c);
  }
}
#define SG_MFMA(a, b, c) sg_mfma<NOMFMA>((a), (b), (c))

__device__ __forceinline__ h8 sg_frag(lds_cptr base, int idx) { return __builtin_bit_cast(h8, base[idx]); }

// One tensor-product block on an operand row X (8 f32x4: s0 s1 x0 x1 y0 y1 z0 z1) with edge / node
// attribute a[3].  ws: this lane's LDS pointer to the operand's scalar k-group of WS (NSB output
// blocks of 16, [mbo][hi|lo][64 lanes]; the vector k-group follows); wt / wv: to its k-group of WT / WV.
//   group 1 (B = s):              S[0..NS) += Ws,   T[0..1] += Wt          NS + 2 accumulators
//   group 2 (B = v.a, v_c):       S[0..NS) += Ws',  V[c][0..1] += Wv       NS + 2 NC accumulators
// NC = vector components carried (2 in 2D: the z component is identically zero).
template <int NC, int NS, int NSB, bool PRIO, bool NOMFMA = false>
__device__ __forceinline__ void sg_operand(lds_cptr ws, lds_cptr wt, lds_cptr wv, const f32x4 (&X)[8],
                                           const float (&a)[3], f32x4 (&S)[4], f32x4 (&T)[2],
                                           f32x4 (&V)[3][2]) {
  {
    h8 sl[NS], tl[2], sh[NS], th[2], bh, bl;
#pragma unroll
    for (int m = 0; m < NS; ++m) sl[m] = sg_frag(ws, (m * 2 + 1) * 64);
#pragma unroll
    for (int m = 0; m < 2; ++m) tl[m] = sg_frag(wt, (m * 2 + 1) * 64);
    SB();
#pragma unroll
    for (int m = 0; m < NS; ++m) sh[m] = sg_frag(ws, (m * 2) * 64);
#pragma unroll
    for (int m = 0; m < 2; ++m) th[m] = sg_frag(wt, (m * 2) * 64);
    SB();
    lb_split8v(X[0], X[1], bh, bl);
    SB();
    if constexpr (PRIO) __builtin_amdgcn_s_setprio(2);
    __builtin_amdgcn_s_waitcnt(LB_WAIT_LGKM(NS + 2));
#pragma unroll
    for (int m = 0; m < NS; ++m) S[m] = SG_MFMA(sl[m], bh, S[m]);
#pragma unroll
    for (int m = 0; m < 2; ++m) T[m] = SG_MFMA(tl[m], bh, T[m]);
    SB();
    __builtin_amdgcn_s_waitcnt(LB_WAIT_LGKM(0));
#pragma unroll
    for (int m = 0; m < NS; ++m) S[m] = SG_MFMA(sh[m], bl, S[m]);
#pragma unroll
    for (int m = 0; m < 2; ++m) T[m] = SG_MFMA(th[m], bl, T[m]);
#pragma unroll
    for (int m = 0; m < NS; ++m) S[m] = SG_MFMA(sh[m], bh, S[m]);
#pragma unroll
    for (int m = 0; m < 2; ++m) T[m] = SG_MFMA(th[m], bh, T[m]);
    SB();
    if constexpr (PRIO) __builtin_amdgcn_s_setprio(0);
  }
  {
    h8 sl[NS], wl[2], sh[NS], wh[2], dh, dl, vh[NC], vl[NC];
#pragma unroll
    for (int m = 0; m < NS; ++m) sl[m] = sg_frag(ws, ((NSB + m) * 2 + 1) * 64);
#pragma unroll
    for (int m = 0; m < 2; ++m) wl[m] = sg_frag(wv, (m * 2 + 1) * 64);
    SB();
#pragma unroll
    for (int m = 0; m < NS; ++m) sh[m] = sg_frag(ws, ((NSB + m) * 2) * 64);
#pragma unroll
    for (int m = 0; m < 2; ++m) wh[m] = sg_frag(wv, (m * 2) * 64);
    SB();
    f32x4 d0 = X[2] * a[0] + X[4] * a[1], d1 = X[3] * a[0] + X[5] * a[1];
    if constexpr (NC == 3) {
      d0 = d0 + X[6] * a[2];
      d1 = d1 + X[7] * a[2];
    }
    lb_split8v(d0, d1, dh, dl);
#pragma unroll
    for (int c = 0; c < NC; ++c) lb_split8v(X[2 + 2 * c], X[3 + 2 * c], vh[c], vl[c]);
    SB();
    if constexpr (PRIO) __builtin_amdgcn_s_setprio(2);
    __builtin_amdgcn_s_waitcnt(LB_WAIT_LGKM(NS + 2));
#pragma unroll
    for (int m = 0; m < NS; ++m) S[m] = SG_MFMA(sl[m], dh, S[m]);
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int m = 0; m < 2; ++m) V[c][m] = SG_MFMA(wl[m], vh[c], V[c][m]);
    SB();
    __builtin_amdgcn_s_waitcnt(LB_WAIT_LGKM(0));
#pragma unroll
    for (int m = 0; m < NS; ++m) S[m] = SG_MFMA(sh[m], dl, S[m]);
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int m = 0; m < 2; ++m) V[c][m] = SG_MFMA(wh[m], vl[c], V[c][m]);
#pragma unroll
    for (int m = 0; m < NS; ++m) S[m] = SG_MFMA(sh[m], dh, S[m]);
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int m = 0; m < 2; ++m) V[c][m] = SG_MFMA(wh[m], vh[c], V[c][m]);
    SB();
    if constexpr (PRIO) __builtin_amdgcn_s_setprio(0);
  }
}

// sigma(x) on z = -log2(e) x (the scale is in the weights): v_exp_f32, v_rcp_f32 per value, the "1 +" packed
__device__ __forceinline__ f32x4 sg_sig4(const f32x4& z) {
  f32x4 e;
#pragma unroll
  for (int j = 0; j < 4; ++j) e[j] = __builtin_amdgcn_exp2f(z[j]);
  const f32x2v one = {1.f, 1.f};
  const f32x2v d0 = lb_lo2(e) + one, d1 = lb_hi2(e) + one;
  return f32x4{__builtin_amdgcn_rcpf(d0[0]), __builtin_amdgcn_rcpf(d0[1]), __builtin_amdgcn_rcpf(d1[0]),
               __builtin_amdgcn_rcpf(d1[1])};
}

// e3nn gate on the accumulators of one block: H_s = z sigma(z) (the silu's constants are folded into the
// consumer's weights, or applied by the caller), H_v[c] = (V[c] + T a_c) sigma(gate); packed fp32 (v_pk_*: one
// issue slot for two values - a single wave issues one VALU instruction per ~5 cycles whatever its width)
template <int NC>
__device__ __forceinline__ void sg_gate(const f32x4 (&S)[4], const f32x4 (&T)[2], const f32x4 (&V)[3][2],
                                        const float (&at)[3], f32x4 (&H)[8]) {
  const f32x4 g0 = sg_sig4(S[2]), g1 = sg_sig4(S[3]);
  const f32x4 s0 = sg_sig4(S[0]), s1 = sg_sig4(S[1]);
  H[0] = lb_cat2(lb_lo2(S[0]) * lb_lo2(s0), lb_hi2(S[0]) * lb_hi2(s0));
  H[1] = lb_cat2(lb_lo2(S[1]) * lb_lo2(s1), lb_hi2(S[1]) * lb_hi2(s1));
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const f32x2v a2 = {at[c], at[c]};
    H[2 + 2 * c] = lb_cat2(__builtin_elementwise_fma(lb_lo2(T[0]), a2, lb_lo2(V[c][0])) * lb_lo2(g0),
                           __builtin_elementwise_fma(lb_hi2(T[0]), a2, lb_hi2(V[c][0])) * lb_hi2(g0));
    H[3 + 2 * c] = lb_cat2(__builtin_elementwise_fma(lb_lo2(T[1]), a2, lb_lo2(V[c][1])) * lb_lo2(g1),
                           __builtin_elementwise_fma(lb_hi2(T[1]), a2, lb_hi2(V[c][1])) * lb_hi2(g1));
  }
}

