// lb_node16s.hip - node MLP (+ residual + next-layer sender/receiver projection), f16x2 arithmetic,
// weights streamed ONCE per compute unit through an LDS ring filled by direct-to-LDS loads.
//
// Reference: GNS._encoder node branch (models/gns.py:65-72), the processor's update_node_features +
// residual (gns.py:103-113,120-122) and the per-node half of the next edge MLP's first Linear
// (W0[:2D] split, lb_gns.hip header).
//
// Round-1 kernel (lb_node16h.hip): 256-thread workgroups of 4 tiles, every workgroup streams all
// three matrices (320 KiB as fp16 hi|lo) through registers into a two-slot LDS ring: 1000 workgroups x
// 320 KiB = 320 MB of L2 -> LDS traffic per launch for 164 MB of node data, and ten
// load -> wait -> ds_write -> barrier steps per 64 nodes.  Measured 73 us = 0.28 of the HBM roofline.
// Here (64 k nodes = 4000 16-node tiles = 15.6 per CU):
//   * workgroups of NW waves, one 16-node tile per wave.  Default NW = 8, two workgroups per CU (<= 128 VGPRs):
//     TGV3D-8k x 8 is one round of 500 workgroups, the weights cross L2 -> LDS twice per CU (164 MB per launch
//     instead of 320 MB).  NW = 16 (LB_NODE_NW=16: one workgroup per CU, 82 MB) measures 4 us slower, NW = 4 is
//     used for small launches so that more CUs take part;
//   * the stream is 10 uniform 32 KiB chunks (2 k-steps x 8 output blocks x hi|lo; the projection is
//     packed as [Ws | Wr] halves so that its chunks have the same shape) landing in a two-slot (NW 16: four-slot)
//     ring via global_load_lds_dwordx4 (no staging registers), one barrier per chunk;
//   * per-wave register diet for four waves per SIMD: the aggregated messages arrive in two halves
//     into the registers the node-latent half just released, the projection runs as two 128-wide
//     halves, the residual re-reads the node row (L2 hit) instead of keeping it for the whole pass;
//   * the inner block loop / split / LayerNorm are the ones of lb_edge16v.hip (lb_f16x2.h).
// Measured 52-54 us per launch; where the time goes: profiles/r02_node16s_ablation.txt, DESIGN.md section 4.4 / profiles/HISTORY.md.
#include <stdlib.h>

#include "lb_f16x2.h"

#define NS_CHUNK 2048  // f32x4 per chunk (32 KiB)

typedef __attribute__((address_space(3))) f32x4* lds_ptr;
typedef const __attribute__((address_space(1))) void* gbl_cvptr;

// aggregated messages of one node, features 16*mb + 4*g .. +3 for mb = MB0 .. MB0+3 (jraph.segment_sum
// output: `agg[r]` when the row sat in one edge tile, else the per-tile partial slots, in tile order)
template <int MB0>
__device__ __forceinline__ void lb_load_agg_half(const lb_node_args& a, int64_t gnode, int g, int k0, int k1,
                                                 f32x4 (&v)[4]) {
  if (!a.fused) {
    const f32x4* gr = reinterpret_cast<const f32x4*>(a.agg) + gnode * 32 + g;
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) v[mb] = gr[4 * (MB0 + mb)];
    // taken delivery of HERE, like the fused branch below does by summing: a load left pending on one side of the branch
    // makes the compiler wait at the join with vmcnt(0), which - the counter is in order - also drains every weight chunk
    // requested since (k_node16q keeps three in flight)
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) asm volatile("" : "+v"(v[mb]));
    return;
  }
  const int t0 = k0 >> a.tile_shift, t1 = (k1 - 1) >> a.tile_shift;
  const bool single = t0 == t1;
  const int nsrc = (k1 <= k0) ? 0 : (single ? 1 : t1 - t0 + 1);
#pragma unroll
  for (int mb = 0; mb < 4; ++mb) v[mb] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int s = 0; __any(s < nsrc); ++s) {
    if (s < nsrc) {
      const int t = t0 + s;
      const float* src = single ? a.agg + gnode * 128
                                : a.part + ((int64_t)t * 2 + (k0 <= (t << a.tile_shift) ? 0 : 1)) * 128;
      const f32x4* s4 = reinterpret_cast<const f32x4*>(src) + g;
#pragma unroll
      for (int mb = 0; mb < 4; ++mb) v[mb] = v[mb] + s4[4 * (MB0 + mb)];
    }
  }
}

// NPA: k-steps (of 32) of input A (encoder features: 1 or 2; node latents: 4); NPB: 4 when the
// aggregated messages are a second input (processor), else 0.  NW: waves (= 16-node tiles) per workgroup.
// NS_SLOTS: ring depth.  (NW 16, 4 slots): one workgroup per CU, weights cross L2 -> LDS once per CU;
// (NW 8, 2 slots): two workgroups per CU running out of phase, so that one's row loads / stores overlap
// the other's MFMA steps (all waves of ONE workgroup move in lock step between the chunk barriers).
template <int NPA, int NPB, bool RESID, bool PROJ, int NW, int NS_SLOTS>
__global__ void __launch_bounds__(NW * 64, NW == 4 ? (NS_SLOTS == 4 ? 1 : 2) : 4)
    k_node16s(lb_node_args a, const f32x4* __restrict__ w0h, const f32x4* __restrict__ w1h,
              const f32x4* __restrict__ wph) {
  __shared__ f32x4 sB[NS_SLOTS][NS_CHUNK];
  __shared__ f32x4 sP[192];  // [0,32) b0, [32,64) b1, [64,96) ln_s, [96,128) ln_o, [128,192) bp
  if (a.ctrl->overflow_step >= 0) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, g = lane >> 4;
  constexpr int NP0 = NPA + NPB;
  constexpr int NCH0 = (NP0 + 1) / 2;
  constexpr int n_chunks = NCH0 + 2 + (PROJ ? 4 : 0);
  static_assert(NPB == 0 || (NPA == 4 && NPB == 4), "processor shape: 4 + 4 k-steps");

  if (tid < 128) {
    const float* src = tid < 32 ? a.b0 : (tid < 64 ? a.b1 : (tid < 96 ? a.ln_s : a.ln_o));
    sP[tid] = reinterpret_cast<const f32x4*>(src)[tid & 31];
  } else if (tid < 192) {
    sP[tid] = (PROJ && a.bp) ? reinterpret_cast<const f32x4*>(a.bp)[tid - 128] : f32x4{0.f, 0.f, 0.f, 0.f};
  }

  // ---- weight stream: chunk c -> (source, valid f32x4); every wave moves pieces wave, wave+NW, ...
  auto issue_chunk = [&](int c) {
    const f32x4* src;
    int nvec = NS_CHUNK;
    if (c < NCH0) {
      src = w0h + (size_t)c * NS_CHUNK;
      if (2 * c + 2 > NP0) nvec = NS_CHUNK / 2;
    } else if (c < NCH0 + 2) {
      src = w1h + (size_t)(c - NCH0) * NS_CHUNK;
    } else {
      src = wph + (size_t)(c - NCH0 - 2) * NS_CHUNK;
    }
    f32x4* slot = sB[c % NS_SLOTS];
#pragma unroll
    for (int i = 0; i < 32 / NW; ++i) {
      const int piece = wave + NW * i;  // 1 KiB per wave instruction
      // inline asm on purpose: through the builtin the compiler knows the instruction writes LDS,
      // cannot tell the ring slots apart and puts s_waitcnt vmcnt(0) in front of the next ds_read -
      // i.e. every step would wait for the refill it has just issued
      if (piece * 64 < nvec) {
        // scalar chunk base + 32-bit lane offset: no 64-bit VGPR address pair per piece
        const uint32_t voff = (uint32_t)(piece * 64 + lane) * 16u;
        const uint32_t lo = (uint32_t)(uintptr_t)(lds_ptr)(slot + piece * 64);
        asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(src), "s"(lo) : "memory");
      }
    }
  };
  // step c: chunk c has landed in every wave and everybody has left chunk c-1.  vmcnt counts in order:
  // chunk c's pieces are older than everything issued after them, i.e. the pieces of the (at most two)
  // later chunks - waiting down to that many outstanding operations is enough and leaves the newest
  // chunks (and nothing else that matters) in flight.  A step then issues its data loads FIRST and the
  // refill of the slot chunk c-1 released LAST, so that a later wait for the data does not have to
  // cover the refill.
  constexpr int DPC = 32 / NW;  // direct-to-LDS instructions per wave and chunk
#define NS_STEP(c)                                                                     \
  do {                                                                                 \
    constexpr int later = ((c) + NS_SLOTS - 2 < n_chunks ? NS_SLOTS - 2 : n_chunks - 1 - (c)); \
    if constexpr (later * DPC == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   \
    if constexpr (later * DPC == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");   \
    if constexpr (later * DPC == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");   \
    if constexpr (later * DPC == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   \
    if constexpr (later * DPC == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   \
    if constexpr (later * DPC == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); \
    if constexpr ((c) == 0) {                                                          \
      __syncthreads(); /* also publishes the sP vectors written above */               \
    } else {                                                                           \
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");                  \
    }                                                                                  \
  } while (0)
#define NS_REFILL(c)                                                           \
  do {                                                                         \
    if constexpr ((c) >= 1 && (c)-1 + NS_SLOTS < n_chunks) issue_chunk((c)-1 + NS_SLOTS); \
  } while (0)
#pragma unroll
  for (int c = 0; c < (n_chunks < NS_SLOTS ? n_chunks : NS_SLOTS); ++c) issue_chunk(c);

  const int64_t row = ((int64_t)blockIdx.x * NW + wave) * 16 + n;
  const bool valid = row < a.n_rows;
  const int64_t rowc = valid ? row : a.n_rows - 1;
  lds_cptr lane_b[NS_SLOTS];
#pragma unroll
  for (int i = 0; i < NS_SLOTS; ++i) lane_b[i] = (lds_cptr)(sB[i] + lane);
  const lds_cptr vecp = (lds_cptr)(sP + g);

  // ---- input A rows (+ the CSR bounds the aggregation loads need, so that they are not a second
  // dependent round trip later)
  f32x4 va[2 * NPA];
  int k0 = 0, k1 = 0;
  {
    const f32x4* xr = reinterpret_cast<const f32x4*>(a.xin) + rowc * (8 * NPA) + g;
#pragma unroll
    for (int mb = 0; mb < 2 * NPA; ++mb) va[mb] = xr[4 * mb];
    if (NPB > 0 && a.fused) {
      k0 = a.row_ptr[rowc];
      k1 = a.row_ptr[rowc + 1];
    }
  }
  const bool probe = wave == 0;  // f16x2 range guard: one tile per workgroup samples its operands
#define NS_BUF(c) lane_b[(c) % NS_SLOTS]
  constexpr int C1 = NCH0;      // first chunk of W1
  constexpr int CP = NCH0 + 2;  // first chunk of the projection
  f32x4 acc[8];
  // ---- GEMM1 over [input A | aggregated messages]
  if constexpr (NPB == 0) {
    NS_STEP(0);
    if (probe) lb_range_probe(a.ctrl, va, 2 * NPA);
#pragma unroll
    for (int mb = 0; mb < 8; ++mb) acc[mb] = vecp[4 * mb];
    lb_gemm16v<false, NPA>(NS_BUF(0), va, acc);
  } else {
    f32x4 h0[4], h1[4];
    NS_STEP(0);
    if (probe) lb_range_probe(a.ctrl, va, 8);
#pragma unroll
    for (int mb = 0; mb < 8; ++mb) acc[mb] = vecp[4 * mb];
    {
      const f32x4 v01[4] = {va[0], va[1], va[2], va[3]};
      lb_gemm16v<false, 2>(NS_BUF(0), v01, acc);
    }
    NS_STEP(1);
    lb_load_agg_half<0>(a, rowc, g, k0, k1, h0);  // into the registers va[0..3] released
    NS_REFILL(1);
    {
      const f32x4 v23[4] = {va[4], va[5], va[6], va[7]};
      lb_gemm16v<false, 2>(NS_BUF(1), v23, acc);
    }
    NS_STEP(2);
    lb_load_agg_half<4>(a, rowc, g, k0, k1, h1);
    NS_REFILL(2);
    lb_gemm16v<false, 2>(NS_BUF(2), h0, acc);
    NS_STEP(3);
    NS_REFILL(3);
    if (probe) {
      lb_range_probe(a.ctrl, h0, 4);
      lb_range_probe(a.ctrl, h1, 4);
    }
    lb_gemm16v<false, 2>(NS_BUF(3), h1, acc);
  }
  if (probe) lb_range_probe(a.ctrl, acc, 8);
  // ---- GEMM2 (ReLU folded into the operand split)
  f32x4 acc2[8];
  NS_STEP(C1);
  NS_REFILL(C1);
#pragma unroll
  for (int mb = 0; mb < 8; ++mb) acc2[mb] = vecp[32 + 4 * mb];
  {
    // per-row TINY test of the hidden row (every tile, one k-group: lb_rows_tiny) - the row the LayerNorm below rescales
    const f32x4 v01[4] = {acc[0], acc[1], acc[2], acc[3]};
    uint32_t or_h = 0;
    lb_gemm16v<true, 2, 1>(NS_BUF(C1), v01, acc2, &or_h);
    if (lb_rows_tiny(or_h) && lane == 0) lb_raise_math(a.ctrl, LB_MATH_TINY);
  }
  NS_STEP(C1 + 1);
  // residual: the node row is read a second time (L2 resident), requested before the refill so that
  // waiting for it later does not wait for the refill
  f32x4 res[8];
  if constexpr (RESID) {
    const f32x4* xr = reinterpret_cast<const f32x4*>(a.xin) + rowc * 32 + g;
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) res[mb] = xr[4 * mb];  // first half now (registers of acc[0..3])
  }
  NS_REFILL(C1 + 1);
  {
    const f32x4 v23[4] = {acc[4], acc[5], acc[6], acc[7]};
    lb_gemm16v<true, 2>(NS_BUF(C1 + 1), v23, acc2);
  }
  if constexpr (RESID) {
    const f32x4* xr = reinterpret_cast<const f32x4*>(a.xin) + rowc * 32 + g;
#pragma unroll
    for (int mb = 4; mb < 8; ++mb) res[mb] = xr[4 * mb];
  }
  // ---- LayerNorm (+ residual)
  f32x4 y[8];
  lb_layernorm16(acc2, vecp + 64, vecp + 96, y, a.ctrl->ln_inv_d, a.ctrl->ln_pad);
  if constexpr (RESID) {
#pragma unroll
    for (int mb = 0; mb < 8; ++mb) y[mb] = lb_pk_add(res[mb], y[mb]);
  }
  if (probe) lb_range_probe(a.ctrl, y, 8);
  if (valid) {
    f32x4* nr = reinterpret_cast<f32x4*>(a.nlat) + row * 32 + g;
#pragma unroll
    for (int mb = 0; mb < 8; ++mb) nr[4 * mb] = y[mb];
  }
  // ---- projection for the next edge MLP: psr = y @ [Ws | Wr] + [0 | b0_next], one 128-wide half at a time
  if constexpr (PROJ) {
    {
      f32x4 accp[8];
      NS_STEP(CP);
      NS_REFILL(CP);
#pragma unroll
      for (int mb = 0; mb < 8; ++mb) accp[mb] = vecp[128 + 4 * mb];
      {
        const f32x4 v01[4] = {y[0], y[1], y[2], y[3]};
        lb_gemm16v<false, 2>(NS_BUF(CP), v01, accp);
      }
      NS_STEP(CP + 1);
      NS_REFILL(CP + 1);
      {
        const f32x4 v23[4] = {y[4], y[5], y[6], y[7]};
        lb_gemm16v<false, 2>(NS_BUF(CP + 1), v23, accp);
      }
      if (valid) {
        f32x4* pr = reinterpret_cast<f32x4*>(a.psr) + row * 64 + g;
#pragma unroll
        for (int mb = 0; mb < 8; ++mb) pr[4 * mb] = accp[mb];
      }
    }
    {
      f32x4 accp[8];
      NS_STEP(CP + 2);
      NS_REFILL(CP + 2);
#pragma unroll
      for (int mb = 0; mb < 8; ++mb) accp[mb] = vecp[128 + 32 + 4 * mb];
      {
        const f32x4 v01[4] = {y[0], y[1], y[2], y[3]};
        lb_gemm16v<false, 2>(NS_BUF(CP + 2), v01, accp);
      }
      NS_STEP(CP + 3);
      NS_REFILL(CP + 3);
      {
        const f32x4 v23[4] = {y[4], y[5], y[6], y[7]};
        lb_gemm16v<false, 2>(NS_BUF(CP + 3), v23, accp);
      }
      if (valid) {
        f32x4* pr = reinterpret_cast<f32x4*>(a.psr) + row * 64 + 32 + g;
#pragma unroll
        for (int mb = 0; mb < 8; ++mb) pr[4 * mb] = accp[mb];
      }
    }
  }
#undef NS_BUF
}
#undef NS_STEP
#undef NS_REFILL

// (round 5: the two measured-slower variants of this kernel left the tree - k_node16s2, two tiles per wave and weight chunk,
// 55 - 57 vs 50.5 us per launch, profiles/r02_node16s_ablation.txt / profiles/HISTORY.md section 4; k_node16q, a four-slot ring of
// 16 KiB chunks with exact in-order vmcnt waits, 51.4 vs 50.6 us, profiles/r04_node_q_ab.txt.  git history has both.)

int lbk_node16s(lb_engine* e, const lb_node_args& a, const float* w0h, const float* w1h, const float* wph2,
                int npa, int npb, bool resid) {
  const f32x4* w0 = reinterpret_cast<const f32x4*>(w0h);
  const f32x4* w1 = reinterpret_cast<const f32x4*>(w1h);
  const f32x4* wp = reinterpret_cast<const f32x4*>(wph2);
  const bool proj = wph2 != nullptr;
  const int64_t tiles = (a.n_rows + 15) / 16;
  // two 8-tile workgroups per CU (2-slot rings); small launches use 4-tile workgroups so that more CUs take part
  // (16-wave workgroups and a deeper ring measured slower: round 2)
  const int nw = tiles < 8 * 256 ? 4 : 8;
  const int nblk = (int)((tiles + nw - 1) / nw);
  dim3 grid(nblk), block(nw * 64);
#define LB_NS(A, B, R, P, W, S) \
  LB_LAUNCH_TIMED(e, (k_node16s<A, B, R, P, W, S>), grid, block, a, w0, w1, wp)
#define LB_NS_W(A, B, R, P)            \
  do {                                 \
    if (nw == 8)                       \
      LB_NS(A, B, R, P, 8, 2);         \
    else                               \
      LB_NS(A, B, R, P, 4, 2);         \
  } while (0)
#define LB_NS_P(A, B, R)               \
  do {                                 \
    if (proj)                          \
      LB_NS_W(A, B, R, true);          \
    else                               \
      LB_NS_W(A, B, R, false);         \
  } while (0)
  if (npa == 4 && npb == 4 && resid)
    LB_NS_P(4, 4, true);
  else if (npa == 1 && npb == 0 && !resid)
    LB_NS_P(1, 0, false);
  else if (npa == 2 && npb == 0 && !resid)
    LB_NS_P(2, 0, false);
  else if (npa == 3 && npb == 0 && !resid)
    LB_NS_P(3, 0, false);
  else if (npa == 4 && npb == 0 && !resid)
    LB_NS_P(4, 0, false);
  else
    return lb_fail(LB_ERR_UNSUPPORTED, "k_node16s<%d,%d,%d> not instantiated", npa, npb, (int)resid);
  LB_HIP(hipGetLastError());
  return LB_OK;
}
