// lb_edge16w.hip - round 5: the processor edge kernel with a DEFERRED EPILOGUE - two tiles in flight per wave.
//
// Same mathematics, layouts, tile walk and summation order as k_edge16v (lb_edge16v.hip; reference:
// GNS._processor update_edge_features + jraph.segment_sum, models/gns.py:86-122) - the results are bit-identical.
// What changes is WHEN a tile's epilogue runs.  k_edge16v executes, per 16-edge tile and wave, 192 MFMAs and then
// ~450 VALU instructions (LayerNorm, residual, the 128-instruction segmented scan, address arithmetic) that depend on
// those MFMAs: in one wave nothing of that overlaps (tools/issue_bench: an MFMA hides two INDEPENDENT VALU
// instructions of its own wave, none that wait for it), and the second wave of the SIMD is mostly waiting for its
// loads while the first computes - rocprofv3: the matrix pipe is busy 37 % of the launch, a SIMD spends ~8.2 k cycles
// per tile, which is the serial sum 192 x 16 (MFMA) + ~740 x 5 (VALU) + LDS reads.  Round 4's VERDICT: "two tiles in
// flight per wave so one tile's LayerNorm / scan / residual VALU fills the other's MFMA slots".
//
// Here the epilogue of tile i-1 (its pre-LayerNorm GEMM output and its latents stay in registers) is cut into ~210
// micro-operations of one or two instructions each (ew_op) and issued, in order, one or two per MFMA slot, inside the two
// GEMMs of tile i (ew_gemm: the block loop of lb_gemm16v with a filler call per slot; every slot is pinned by a
// sched_barrier).  The latents' stores go out in the first GEMM, the aggregates' in the second; both are branch-free
// (the aggregates through raw-buffer stores into the agg | part allocation with the offset pushed out of range where a
// lane has nothing to write), so the compiler's vmcnt bookkeeping stays exact and nothing in the GEMMs waits for a store.
// Two register sets (S0 / S1) alternate as "current" and "previous" tile: the loop body exists twice, no copies.
// The first tile of a wave runs its GEMMs without fillers, the last tile's epilogue runs alone after the loop.
#include <stdlib.h>

#include <type_traits>

#include "lb_f16x2.h"

typedef uint32_t u32x4w __attribute__((ext_vector_type(4)));

// uniform base + 32-bit byte offset: the compiler keeps the base in an SGPR pair (global_load ... v_off, s[base:base+1])
// instead of a 64-bit address per lane - this kernel has no registers to spare (the launcher checks that the buffers
// are below 4 GiB)
__device__ __forceinline__ const f32x4* ew_at(const void* base, uint32_t off) {
  return reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(base) + off);
}
__device__ __forceinline__ f32x4* ew_at(void* base, uint32_t off) {
  return reinterpret_cast<f32x4*>(reinterpret_cast<char*>(base) + off);
}

struct ew_tile {   // what a tile leaves behind for its deferred epilogue
  f32x4 a[8];      // GEMM2 output (pre-LayerNorm) -> normalised -> scanned (C layout: lane (n, g), features 16mb + 4g + j)
  f32x4 ve[8];     // the tile's edge latents (residual)
  int t;           // tile index
  int r_cur;       // receiver of row n
  int rb;          // receivers of the edges just before / after the tile (lb_edge_probe)
  uint32_t or_e, or_h;  // range guard: OR of the fp16 `hi` patterns of the two GEMM operands (GUARD 1)
};

struct ew_ctx {    // wave constants of the epilogue
  lds_cptr vecb;   // this lane's LDS pointer to b1 | ln scale | ln offset
  float ln_inv_d, ln_pad;
  int E, n, g, lane;
  float* elat_out;
  __amdgpu_buffer_rsrc_t out_rs;  // agg | part
  uint32_t part_off;              // byte offset of part inside that allocation
};

struct ew_tmp {    // values that travel from one micro-operation to a later one
  f32x2v tmp, s2, v2, m2, r2, dl, dh, tlo, thi, ylo, olo, ohi;
  float sm, mean, vs, rs;
  f32x4 scn, ofn, scc, ofc;
  int rr, segstart;
  bool valid, head, tail;
  unsigned H;
  float m1, m2f, m4, m8;
  uint32_t ew;
  uint32_t off;
};

// sum over the four lanes {n, n + 16, n + 32, n + 48} of a row, in the order (x0 + x1) + (x2 + x3) of the two
// __shfl_xor steps of lb_layernorm16, on gfx950's row swaps (no LDS round trip)
__device__ __forceinline__ float ew_rowsum4(float x) {
  float p = x, q = x;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(p), "+v"(q));
  p = q = p + q;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(p), "+v"(q));
  return p + q;
}


// Packed (v_pk_*_f32, one instruction per pair) or scalar (two instructions) arithmetic of the LayerNorm / residual
// micro-operations: VERDICT r05 item 2 - MI355X_MICROARCH.md prices a packed-f32 filler beside MFMAs at +22..26 cycles per gap
// against its two scalar instructions.  Measured (tools/issue_bench_pk, tools/edge_ab built with -DEW_SCALAR_LN=1;
// profiles/r06_packed_vs_scalar.txt); the default is what won.
#ifndef EW_SCALAR_LN
#define EW_SCALAR_LN 0
#endif
__device__ __forceinline__ f32x2v ew_add2(f32x2v a, f32x2v b) {
#if EW_SCALAR_LN
  return f32x2v{a[0] + b[0], a[1] + b[1]};
#else
  return a + b;
#endif
}
__device__ __forceinline__ f32x2v ew_sub2(f32x2v a, f32x2v b) {
#if EW_SCALAR_LN
  return f32x2v{a[0] - b[0], a[1] - b[1]};
#else
  return a - b;
#endif
}
__device__ __forceinline__ f32x2v ew_mul2(f32x2v a, f32x2v b) {
#if EW_SCALAR_LN
  return f32x2v{a[0] * b[0], a[1] * b[1]};
#else
  return a * b;
#endif
}
__device__ __forceinline__ f32x2v ew_fma2(f32x2v a, f32x2v b, f32x2v c) {
#if EW_SCALAR_LN
  return f32x2v{__builtin_fmaf(a[0], b[0], c[0]), __builtin_fmaf(a[1], b[1], c[1])};
#else
  return __builtin_elementwise_fma(a, b, c);
#endif
}
template <bool SKIP, int GUARD>
struct ew_plan {  // micro-operation ranges of the deferred epilogue
  static constexpr int A0 = 0;                       // 16: row sums
  static constexpr int A1 = A0 + 16;                 // 3: cross-lane sum, mean
  static constexpr int B0 = A1 + 3;                  // 32: centre, squares
  static constexpr int B1 = B0 + 32;                 // 3: variance, 1 / sigma (+ the first scale / offset reads)
  static constexpr int C0 = B1 + 3;                  // 48: scale, offset
  static constexpr int D0 = C0 + 48;                 // 25: residual + store of the latents (not in the last layer)
  static constexpr int E0 = D0 + (SKIP ? 0 : 25);    // 6: segment structure
  static constexpr int F0 = E0 + 6;                  // 64: segmented scan, two registers per micro-operation
  static constexpr int G0 = F0 + 64;                 // 9: where the sums go + 8 stores
  static constexpr int H0 = G0 + 9;                  // range guard
  static constexpr int M = H0 + (GUARD ? 1 : 0);
};

// y += row_shr:SHR(y) * m for TWO registers behind an s_nop.  Inline asm is invisible to the hazard recogniser: between
// MFMAs the register allocator places live-range-split copies (v_mov) directly in front of such a statement - a VALU write
// one instruction ahead of a DPP read of the same register, two wait states short (a first version without the s_nop
// got every second tile of a wave wrong).  The compiler's own DPP builtin is not an alternative: v_mov_b32_dpp + v_fma,
// the DPP-combine pass does not fold them here (256 instead of 128 instructions per tile).
#define EW_DPP2(SHR, Y0, Y1, MSK)                                                                     \
  asm volatile("s_nop 1\n\t"                                                                          \
               "v_fmac_f32_dpp %0, %0, %2 row_shr:" #SHR " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
               "v_fmac_f32_dpp %1, %1, %2 row_shr:" #SHR " row_mask:0xf bank_mask:0xf bound_ctrl:1"      \
               : "+v"(Y0), "+v"(Y1)                                                                   \
               : "v"(MSK))

// Micro-operation k of the epilogue of tile P.  k is a compile-time constant at every call site (after unrolling).
// PIPE: the LayerNorm scale / offset rows are read from LDS one block ahead (as fillers between MFMAs); eager execution
// (PIPE false) reads them where they are used and leaves the scheduling to the compiler (8 registers fewer).
template <bool SKIP, bool NT, int GUARD, bool PIPE = true>
__device__ __forceinline__ void ew_op(int k, ew_tile& P, ew_tmp& T, const ew_ctx& C, int& guard_tiny) {
  using PL = ew_plan<SKIP, GUARD>;
  if (k < PL::A1) {  // s2 = sum over the lane's 32 features, two at a time
    const int mb = k >> 1;
    if (!(k & 1))
      T.tmp = ew_add2(lb_lo2(P.a[mb]), lb_hi2(P.a[mb]));
    else
      T.s2 = mb == 0 ? T.tmp : ew_add2(T.s2, T.tmp);
  } else if (k < PL::B0) {
    const int j = k - PL::A1;
    if (j == 0)
      T.sm = T.s2[0] + T.s2[1];
    else if (j == 1)
      T.sm = ew_rowsum4(T.sm);
    else {
      T.mean = T.sm * C.ln_inv_d;
      T.m2 = f32x2v{T.mean, T.mean};
    }
  } else if (k < PL::B1) {
    const int j = k - PL::B0, mb = j >> 2, sub = j & 3;
    if (sub == 0)
      T.dl = ew_sub2(lb_lo2(P.a[mb]), T.m2);
    else if (sub == 1)
      T.dh = ew_sub2(lb_hi2(P.a[mb]), T.m2);
    else if (sub == 2)
      T.v2 = mb == 0 ? ew_mul2(T.dl, T.dl) : ew_fma2(T.dl, T.dl, T.v2);
    else {
      T.v2 = ew_fma2(T.dh, T.dh, T.v2);
      P.a[mb] = lb_cat2(T.dl, T.dh);
    }
  } else if (k < PL::C0) {
    const int j = k - PL::B1;
    if (j == 0) {
      T.vs = T.v2[0] + T.v2[1];
      if constexpr (PIPE) T.scn = C.vecb[32];
    } else if (j == 1) {
      T.vs = ew_rowsum4(T.vs);
      if constexpr (PIPE) T.ofn = C.vecb[64];
    } else {
      // a latent narrower than 128 is zero-padded: the padded entries each contributed mean^2
      T.rs = 1.0f / sqrtf(fmaxf(T.vs - C.ln_pad * (T.mean * T.mean), 0.f) * C.ln_inv_d + 1e-5f);
      T.r2 = f32x2v{T.rs, T.rs};
    }
  } else if (k < PL::D0) {
    const int j = k - PL::C0, mb = j / 6, sub = j % 6;
    if (sub == 0) {
      if constexpr (PIPE) {
        T.scc = T.scn;
        if (mb < 7) T.scn = C.vecb[32 + 4 * (mb + 1)];  // the next block's scale row, one block ahead
      } else {
        T.scc = C.vecb[32 + 4 * mb];
      }
    } else if (sub == 1) {
      if constexpr (PIPE) {
        T.ofc = T.ofn;
        if (mb < 7) T.ofn = C.vecb[64 + 4 * (mb + 1)];
      } else {
        T.ofc = C.vecb[64 + 4 * mb];
      }
    } else if (sub == 2)
      T.tlo = ew_mul2(lb_lo2(T.scc), T.r2);
    else if (sub == 3)
      T.thi = ew_mul2(lb_hi2(T.scc), T.r2);
    else if (sub == 4)
      T.ylo = ew_fma2(T.tlo, lb_lo2(P.a[mb]), lb_lo2(T.ofc));
    else
      P.a[mb] = lb_cat2(T.ylo, ew_fma2(T.thi, lb_hi2(P.a[mb]), lb_hi2(T.ofc)));
  } else if (!SKIP && k < PL::E0) {
    const int j = k - PL::D0;
    if (j == 0) {
      T.ew = (uint32_t)P.t * 8192u + (uint32_t)C.lane * 16u;
    } else {
      const int q = j - 1, mb = q / 3, sub = q % 3;
      if (sub == 0)
        T.olo = ew_add2(lb_lo2(P.ve[mb]), lb_lo2(P.a[mb]));
      else if (sub == 1)
        T.ohi = ew_add2(lb_hi2(P.ve[mb]), lb_hi2(P.a[mb]));
      else {
        const f32x4 o = lb_cat2(T.olo, T.ohi);
        if constexpr (NT)
          __builtin_nontemporal_store(o, ew_at(C.elat_out, T.ew) + 64 * mb);
        else
          ew_at(C.elat_out, T.ew)[64 * mb] = o;
      }
    }
  } else if (k < PL::F0) {  // the receiver segments inside the tile (rows past the end of the list: copies of the last
    const int j = k - PL::E0, n = C.n;  // edge - finite, each its own segment, above every valid lane, never stored)
    if (j == 0) {
      T.valid = P.t * 16 + n < C.E;
      T.rr = T.valid ? P.r_cur : (-1 - n);
    } else if (j == 1) {
      const int r_prev = __builtin_amdgcn_update_dpp(-2, T.rr, 0x111, 0xF, 0xF, false);
      T.head = (n == 0) || (T.rr != r_prev);
    } else if (j == 2) {
      T.H = (unsigned)(__ballot(T.head) & 0xffffull);
      T.segstart = 31 - __clz(T.H & ((2u << n) - 1u));
    } else if (j == 3) {
      T.tail = (n == 15) || ((T.H >> (n + 1)) & 1u);
    } else if (j == 4) {
      T.m1 = (n >= 1 && T.segstart <= n - 1) ? 1.f : 0.f;
      T.m2f = (n >= 2 && T.segstart <= n - 2) ? 1.f : 0.f;
    } else {
      T.m4 = (n >= 4 && T.segstart <= n - 4) ? 1.f : 0.f;
      T.m8 = (n >= 8 && T.segstart <= n - 8) ? 1.f : 0.f;
    }
  } else if (k < PL::G0) {
    // fused jraph.segment_sum: segmented Hillis-Steele scan inside each 16-lane DPP row, x += row_shr:s(x) * m_s, in
    // lb_scan8's order (eight registers per step: a register is read through DPP eight instructions after its write)
    const int j = k - PL::F0, grp = j >> 4, step = (j >> 2) & 3, r = 2 * (j & 3) + 8 * grp;
    if constexpr (!PIPE) {  // executed in one piece (not between MFMAs): eight registers per asm block, one s_nop - lb_scan8
      if ((j & 15) == 0) lb_scan8(P.a[2 * grp], P.a[2 * grp + 1], T.m1, T.m2f, T.m4, T.m8);
    } else if (step == 0)
      EW_DPP2(1, P.a[r >> 2][r & 3], P.a[(r + 1) >> 2][(r + 1) & 3], T.m1);
    else if (step == 1)
      EW_DPP2(2, P.a[r >> 2][r & 3], P.a[(r + 1) >> 2][(r + 1) & 3], T.m2f);
    else if (step == 2)
      EW_DPP2(4, P.a[r >> 2][r & 3], P.a[(r + 1) >> 2][(r + 1) & 3], T.m4);
    else
      EW_DPP2(8, P.a[r >> 2][r & 3], P.a[(r + 1) >> 2][(r + 1) & 3], T.m8);
  } else if (k < PL::H0) {
    const int j = k - PL::G0;
    if (j == 0) {  // rows inside one tile go to agg[r], the <= 2 segments the tile boundary cuts to part[tile][slot]
      int slot01;
      const bool complete = lb_seg_complete(P.rb, T.rr, T.segstart, C.n, P.t, C.E, slot01);
      const uint32_t row_off =
          complete ? (uint32_t)T.rr * 512u : C.part_off + ((uint32_t)P.t * 2u + (uint32_t)slot01) * 512u;
      T.off = (T.tail && T.valid) ? row_off + (uint32_t)C.g * 16u : 0x80000000u;  // out of range: dropped
    } else {
      const int mb = j - 1;
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4w, P.a[mb]), C.out_rs, (int)T.off, 64 * mb, 0);
    }
  } else if (GUARD != 0 && k < PL::M) {
    guard_tiny |= (int)lb_rows_tiny(P.or_e) | (int)lb_rows_tiny(P.or_h);
  }
}

// deferred micro-operations [lo, hi) of MFMA slot s (192 slots per tile: GEMM1 0-95, GEMM2 96-191), spread evenly
template <int DF, int M>
__device__ __forceinline__ constexpr int ew_slot_lo(int s) { return DF + (s * (M - DF)) / 192; }

// acc[0..7] += W^T * v over four k-steps of 32 (f16x2: lo*hi + hi*lo + hi*hi): the block loop of lb_gemm16v
// (phase-pipelined LDS reads, the next operand's split one instruction per slot) + fill(slot) in every MFMA slot.
template <bool RELU, int GUARD, int S0, typename F>
__device__ __forceinline__ void ew_gemm(lds_cptr wbase, const f32x4 (&v)[8], f32x4 (&acc)[8], uint32_t* orv, F&& fill) {
  constexpr int NP = 4;
  auto frag = [&](int p, int mbo, int part) -> h8 {
    return __builtin_bit_cast(h8, wbase[((p * 8 + mbo) * 2 + part) * 64]);
  };
  h8 X[4], Y[4];
  h8 bh, bl;
  lb_split_regs R;
#pragma unroll
  for (int c = 0; c < 4; ++c) X[c] = frag(0, c, 1);
#pragma unroll
  for (int c = 0; c < 4; ++c) Y[c] = frag(0, c, 0);
  {
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
      const u32x4w u = __builtin_bit_cast(u32x4w, bh);
      *orv |= (u[0] | u[1]) | (u[2] | u[3]);
    }
  }
  SB();
#pragma unroll
  for (int blk = 0; blk < 2 * NP; ++blk) {
    const int p = blk >> 1, q = blk & 1;
    const int np = (blk + 1) >> 1, nq = (blk + 1) & 1;
    const bool fs = p < NP - 1;
    const int s0 = 12 * q, sl = S0 + 12 * blk;
    f32x4* a4 = &acc[4 * q];
    __builtin_amdgcn_s_waitcnt(LB_WAIT_LGKM(4));
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      a4[c] = MFMA16H(X[c], bh, a4[c]);
      if (fs) lb_split_slot<RELU, GUARD>(s0 + c, v[2 * p + 2], v[2 * p + 3], R, orv);
      fill(sl + c);
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
      if (fs) lb_split_slot<RELU, GUARD>(s0 + 4 + c, v[2 * p + 2], v[2 * p + 3], R, orv);
      fill(sl + 4 + c);
      SB();
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      a4[c] = MFMA16H(Y[c], bh, a4[c]);
      if (fs) lb_split_slot<RELU, GUARD>(s0 + 8 + c, v[2 * p + 2], v[2 * p + 3], R, orv);
      fill(sl + 8 + c);
      SB();
    }
    if (blk < 2 * NP - 1) {
#pragma unroll
      for (int c = 0; c < 4; ++c) Y[c] = frag(np, 4 * nq + c, 0);
    }
    if (q == 1 && fs) {
      bh = __builtin_bit_cast(h8, u32x4w{R.H[0], R.H[1], R.H[2], R.H[3]});
      bl = __builtin_bit_cast(h8, u32x4w{R.L[0], R.L[1], R.L[2], R.L[3]});
    }
    SB();
  }
}

// SKIP: last processor layer (the updated edge latents have no reader); NT: nontemporal edge-latent streams;
// GUARD: 0 sampled probe only, 1 + per-row TINY test of every tile (the default in guarded mode); TICKET: the waves of a
// workgroup draw their tiles from an LDS counter (single trajectories).
// DEFER: what rides in the next tile's MFMA slots.  0: nothing (the epilogue behind the GEMMs, as k_edge16v); 1: the
// segment structure, the scan, the aggregate stores and the range-guard test (80 micro-operations) - LayerNorm, residual
// and the latents' stores run right behind the tile's own GEMMs; 2: everything (the tile's latents travel with it).
// PF: when a tile's loads are issued.
//   0: at the top of the tile (k_edge16v's schedule; the sender rows are needed only after GEMM1).
//   2 (DEEP): the edge latents - the HBM stream - a WHOLE TILE ahead (at the top of the tile before, behind nothing but
//      that tile's gathers), the gathered projections, the next indices and the probe in the tail of the tile before,
//      AHEAD of its stores.  gfx9 has one in-order vmcnt for loads and stores: a wait for loads issued behind stores
//      drains the stores; here no wait ever names an operation younger than a store it does not need, and every count
//      is exact (the prefetch sits under a uniform branch, its count only ever makes a wait later than necessary).
//      Registers: the next tile's latents (32) live through both GEMMs - 2 waves per SIMD still fit without DEFER 2's
//      travelling latents (storing variant: DEFER <= 1).
// ABL (tools/edge_ab.hip only): 1 the deferred part in one piece ahead of the GEMMs (debug), 2 GEMM-phase priority.
template <bool SKIP, bool NT, int GUARD, bool TICKET, int DEFER = 1, int ABL = 0, int WPS = 2, int PF = 0, int WALK = 0>
__global__ void __launch_bounds__(WPS * 256, 1) k_edge16w(lb_edge16_args a) {
  static_assert(PF == 0 || PF == 2, "PF: 0 or 2");
  static_assert(!(PF == 2 && DEFER == 2 && !SKIP), "DEEP prefetch overwrites the previous tile's latents");
  constexpr int THREADS = WPS * 256, WAVES = WPS * 4;
  constexpr int NW0 = 4096;
  using PL = ew_plan<SKIP, GUARD>;
  constexpr int DF = DEFER == 2 ? 0 : (DEFER == 1 ? PL::E0 : PL::M);  // first deferred micro-operation
  __shared__ f32x4 sW[NW0 + 4096 + 96];
  __shared__ int s_ticket;
  // prologue as in k_edge16v: control block, weight loads into registers, the first tile's indices, then LDS
  const int poisoned = a.ctrl->overflow_step;
  const int E = a.ctrl->n_edges_total;
  const float ln_inv_d = a.ctrl->ln_inv_d, ln_pad = a.ctrl->ln_pad;
  const int tid = threadIdx.x;
  constexpr int NST = (NW0 + 4096 + THREADS - 1) / THREADS;
  f32x4 st[NST];
  {
    const f32x4* g0 = reinterpret_cast<const f32x4*>(a.w0p);
    const f32x4* g1 = reinterpret_cast<const f32x4*>(a.w1p);
#pragma unroll
    for (int k = 0; k < NST; ++k) {
      const int i = tid + k * THREADS;
      st[k] = i < NW0 ? g0[i] : g1[(i < NW0 + 4096 ? i : NW0 + 4095) - NW0];
    }
  }
  const int ntiles = (E + 15) >> 4;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, g = lane >> 4;
  const int xcd = blockIdx.x & 7;
  int stride = (gridDim.x >> 3) * WAVES;
  int t_lo = (int)(((int64_t)ntiles * xcd) >> 3), t_hi = (int)(((int64_t)ntiles * (xcd + 1)) >> 3);
  int t = t_lo + (blockIdx.x >> 3) * WAVES + wave;
  if constexpr (TICKET) {
    stride = gridDim.x >> 3;
    t_lo += (blockIdx.x >> 3);
    t = t_lo + wave * stride;
  }
  auto rowc_of = [&](int tt) -> int64_t {
    const int row = tt * 16 + n;
    return row < E ? row : (E > 0 ? E - 1 : 0);
  };
  // a.reverse: the XCD's range [t_lo0, t_hi) is walked from its end (the walk position tw still runs upwards)
  const int t_lo0 = (int)(((int64_t)ntiles * xcd) >> 3);
  const int rev_sum = a.reverse ? t_lo0 + t_hi - 1 : 0;
  if constexpr (WALK == 1) {  // (experiment) every wave owns a CONTIGUOUS chunk of its XCD's range instead of a stride
    const int slot = (blockIdx.x >> 3) * WAVES + wave, nslot = (gridDim.x >> 3) * WAVES, ntx = t_hi - t_lo0;
    t = t_lo0 + (int)(((int64_t)ntx * slot) / nslot);
    t_hi = t_lo0 + (int)(((int64_t)ntx * (slot + 1)) / nslot);
    stride = 1;
  }
  auto tmap = [&](int tw) -> int { return a.reverse ? rev_sum - tw : tw; };
  int s_c = 0, r_c = 0;
  if (t < t_hi) {
    const int64_t rc = rowc_of(tmap(t));
    s_c = a.senders[rc];
    r_c = a.receivers[rc];
  }
  if (poisoned >= 0) return;
  {
#pragma unroll
    for (int k = 0; k < NST; ++k) {
      const int i = tid + k * THREADS;
      if (i < NW0 + 4096) sW[i] = st[k];
    }
    if (tid < 96) {
      const float* src = tid < 32 ? a.b1 : (tid < 64 ? a.ln_s : a.ln_o);
      sW[NW0 + 4096 + tid] = reinterpret_cast<const f32x4*>(src)[tid & 31];
    }
    if (TICKET && tid == 0) s_ticket = WAVES;
  }
  __syncthreads();
  if (t >= t_hi) return;
  uint32_t off0 = (uint32_t)(uintptr_t)(lds_cptr)(sW + lane);
  uint32_t off1 = (uint32_t)(uintptr_t)(lds_cptr)(sW + NW0 + lane);
  uint32_t off2 = (uint32_t)(uintptr_t)(lds_cptr)(sW + NW0 + 4096 + g);
  asm volatile("" : "+v"(off0), "+v"(off1), "+v"(off2));
  const lds_cptr w0b = (lds_cptr)(uintptr_t)off0, w1b = (lds_cptr)(uintptr_t)off1, vecb = (lds_cptr)(uintptr_t)off2;
  const int t_last = TICKET ? t_hi - 1 : t + ((t_hi - 1 - t) / stride) * stride;
  ew_ctx C;
  C.vecb = vecb;
  C.ln_inv_d = ln_inv_d;
  C.ln_pad = ln_pad;
  C.E = E;
  C.n = n;
  C.g = g;
  C.lane = lane;
  C.elat_out = a.elat_out ? a.elat_out : a.elat;
  C.out_rs = __builtin_amdgcn_make_buffer_rsrc(a.agg, 0, (int)a.aggpart_bytes, 0x00020000);
  C.part_off = (uint32_t)((const char*)a.part - (const char*)a.agg);
  asm volatile("" : "+v"(s_c), "+v"(r_c));
  int guard_tiny = 0;
  ew_tile S0, S1;
  ew_tmp T;
  f32x4 acc[8], p0[8];  // gathered projections of the tile whose loads are in flight (receiver rows / sender rows)
  int t_next = 0, s_cur = 0;

  auto draw_next = [&](int tw) -> int {  // the walk position after tw
    if constexpr (TICKET) {
      int k = 0;
      if (lane == 0) k = atomicAdd(&s_ticket, 1);
      return t_lo + __builtin_amdgcn_readfirstlane(k) * stride;
    } else {
      return tw + stride;
    }
  };
  auto issue_ve = [&](ew_tile& X, int tw) {  // the edge latents of walk position tw: the HBM stream
    const f32x4* er = ew_at(a.elat, (uint32_t)tmap(tw) * 8192u + (uint32_t)lane * 16u);
#pragma unroll
    for (int mb = 0; mb < 8; ++mb) X.ve[mb] = NT ? __builtin_nontemporal_load(&er[64 * mb]) : er[64 * mb];
  };
  // everything else tile tw needs (its indices s_c / r_c are here): the indices of the walk position after it (tw_after),
  // its probe, the receiver rows (GEMM1's start value) and - WITH_PS - the sender rows
  auto issue_rest = [&](ew_tile& X, int tw, int tw_after, auto with_ps) {
    const int tt = tmap(tw);
    X.t = tt;
    X.r_cur = r_c;
    s_cur = s_c;
    const f32x4* pr = ew_at(a.psr, (uint32_t)r_c * 1024u + 512u + (uint32_t)g * 16u);
    const f32x4* ps = ew_at(a.psr, (uint32_t)s_c * 1024u + (uint32_t)g * 16u);
    const uint32_t rn = (uint32_t)rowc_of(tmap(min(tw_after, t_last))) * 4u;
    s_c = *reinterpret_cast<const int32_t*>(reinterpret_cast<const char*>(a.senders) + rn);
    r_c = *reinterpret_cast<const int32_t*>(reinterpret_cast<const char*>(a.receivers) + rn);
    int lane_o = lane;  // (opaque copy: the probe's per-lane constant is recomputed, not kept in a register across the tile)
    asm volatile("" : "+v"(lane_o));
    X.rb = lb_edge_probe(a.receivers, tt, lane_o, E);
#pragma unroll
    for (int mb = 0; mb < 8; ++mb) {
      acc[mb] = pr[4 * mb];
      if constexpr (decltype(with_ps)::value) p0[mb] = ps[4 * mb];
    }
  };
  std::true_type yes;
  std::false_type no;
  // One tile.  The SENDER projections are needed only after GEMM1: the start value of GEMM1 is the receiver row alone,
  // h = (p_r + W0 e) + p_s (k_edge16v: (p_s + p_r) + W0 e; fp32 re-association, same tolerance class).
  // GEMM1 / GEMM2 carry the deferred part of the PREVIOUS tile's epilogue in their MFMA slots (HAS_PREV).
  auto body = [&](auto has_prev, auto first, ew_tile& cur, ew_tile& prev) {
    constexpr bool HAS_PREV = decltype(has_prev)::value, FIRST = decltype(first)::value;
    if constexpr (PF == 0) {
      t_next = draw_next(t);
      issue_ve(cur, t);
      issue_rest(cur, t, t_next, no);
      {
        const f32x4* ps = ew_at(a.psr, (uint32_t)s_cur * 1024u + (uint32_t)g * 16u);
#pragma unroll
        for (int mb = 0; mb < 8; ++mb) p0[mb] = ps[4 * mb];
      }
      // the next tile's indices and the probe are taken delivery of HERE, with the tile's own operands and before the
      // first store: a later wait for them would be a drain of those stores
      asm volatile("" : "+v"(s_c), "+v"(r_c), "+v"(cur.rb));
    } else {
      // DEEP: this tile's operands are in flight or here; the NEXT tile's latents go out now, a whole tile ahead
      if (t_next < t_hi) issue_ve(prev, t_next);
    }
    f32x4 h[8];
#pragma unroll
    for (int mb = 0; mb < 8; ++mb) h[mb] = acc[mb];
    if constexpr (FIRST) lb_range_probe(a.ctrl, cur.ve, 8);  // f16x2 range guard, first tile of every wave
    if constexpr (ABL & 2) __builtin_amdgcn_s_setprio(2);
    cur.or_e = 0;
    cur.or_h = 0;
    if constexpr (HAS_PREV && (ABL & 1)) {  // (debug) the deferred part in one piece ahead of the GEMMs
#pragma unroll
      for (int k = DF; k < PL::M; ++k) ew_op<SKIP, NT, GUARD, false>(k, prev, T, C, guard_tiny);
    }
    auto fill = [&](int s) {
      if constexpr (HAS_PREV && DF < PL::M && !(ABL & 1)) {
        const int lo = ew_slot_lo<DF, PL::M>(s), hi = ew_slot_lo<DF, PL::M>(s + 1);
#pragma unroll
        for (int u = 0; u < 3; ++u)
          if (lo + u < hi) ew_op<SKIP, NT, GUARD>(lo + u, prev, T, C, guard_tiny);
      }
    };
    ew_gemm<false, GUARD, 0>(w0b, cur.ve, h, &cur.or_e, fill);
#pragma unroll
    for (int mb = 0; mb < 8; ++mb) h[mb] = lb_pk_add(h[mb], p0[mb]);
    if constexpr (FIRST) lb_range_probe(a.ctrl, h, 8);
#pragma unroll
    for (int mb = 0; mb < 8; ++mb) cur.a[mb] = vecb[4 * mb];
    ew_gemm<true, GUARD, 96>(w1b, h, cur.a, &cur.or_h, fill);
    if constexpr (ABL & 2) __builtin_amdgcn_s_setprio(0);
    if constexpr (PF == 2) {
      // the gathered rows, the indices after next and the probe of the NEXT tile: ahead of this tile's stores
      // (a uniform branch around LOADS that are followed by unconditional stores keeps the compiler's vmcnt exact)
      const int t_nn = t_next < t_hi ? draw_next(t_next) : t_next;
      if (t_next < t_hi) issue_rest(prev, t_next, t_nn, yes);
      t = t_next;
      t_next = t_nn;
    } else {
      t = t_next;
    }
#pragma unroll
    for (int k = 0; k < DF; ++k) ew_op<SKIP, NT, GUARD, false>(k, cur, T, C, guard_tiny);
  };
  auto drain = [&](ew_tile& last) {
#pragma unroll
    for (int k = DF; k < PL::M; ++k) ew_op<SKIP, NT, GUARD, false>(k, last, T, C, guard_tiny);
  };
  if constexpr (PF == 2) {  // the first tile's loads (its indices are here), the second tile's walk position
    t_next = draw_next(t);
    issue_ve(S0, t);
    issue_rest(S0, t, t_next, yes);
  }
  body(no, yes, S0, S1);
  for (;;) {
    if (t >= t_hi) {
      drain(S0);
      break;
    }
    body(yes, no, S1, S0);
    if (t >= t_hi) {
      drain(S1);
      break;
    }
    body(yes, no, S0, S1);
  }
  if (guard_tiny && lane == 0) lb_raise_math(a.ctrl, LB_MATH_TINY);
}

// Launcher: the grid / ticket / nontemporal / guard selection of lbk_edge16v.  Engines in LB_GUARD=full mode (every
// k-group of every tile range-tested), aggregate buffers beyond 2 GiB and edge buffers beyond 4 GiB stay on k_edge16v
// (lb_gns.hip: lb_use_edge_w).  Variants (tools/edge_ab, profiles/r05_edge_ab.txt; TGV3D-8k x 8 shapes, E = 1.037 M):
//   storing layers: loads at the top, nothing deferred      252 us against k_edge16v's 260 (fewer instructions: the sender
//                   rows join after GEMM1, row sums on permlane swaps, branch-free aggregate stores);
//                   deferred scan 255, deep prefetch 262, alternating walk direction +-0 - the launch is POWER bound
//                   (profiles/r05_clock_probe.txt: 1400 W, shader clock 1.7 GHz instead of 2.4), schedules do not matter;
//   last layer (no store): everything deferred              215 - 219 us against 227.
#ifndef EW_NO_LAUNCHER
int lbk_edge16w(lb_engine* e, const lb_edge16_args& a) {
  // one trajectory (< 12288 capacity tiles): LDS tile tickets, plain accesses; batches: static walk, nontemporal streams
  // (lbk_edge16v has the measurements)
  const int64_t tiles_cap = ((int64_t)e->e_cap * e->g.B + 15) / 16;
  const bool small = tiles_cap < 12288;
  int64_t g = (tiles_cap + 7) / 8;
  g = (g + 7) / 8 * 8;
  const int grid = (int)(g < 8 ? 8 : (g > 256 ? 256 : g));
  const bool guard = e->math_auto && !e->guard_sampled;
#define LB_EW4(SK, NT_, GU, TK) \
  LB_LAUNCH_TIMED(e, (k_edge16w<SK, NT_, GU, TK, (SK ? 2 : 0)>), dim3(grid), dim3(512), a)
#define LB_EW2(SK, NT_, TK)  \
  do {                       \
    if (guard)               \
      LB_EW4(SK, NT_, 1, TK);\
    else                     \
      LB_EW4(SK, NT_, 0, TK);\
  } while (0)
#define LB_EW1(SK)              \
  do {                          \
    if (small)                  \
      LB_EW2(SK, false, true);  \
    else                        \
      LB_EW2(SK, true, false);  \
  } while (0)
  if (a.skip_elat_store)
    LB_EW1(true);
  else
    LB_EW1(false);
#undef LB_EW1
#undef LB_EW2
#undef LB_EW4
  LB_HIP(hipGetLastError());
  return LB_OK;
}
#endif
