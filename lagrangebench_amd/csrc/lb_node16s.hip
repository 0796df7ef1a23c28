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
// Measured 52-54 us per launch; where the time goes: profiles/r02_node16s_ablation.txt, DESIGN.md section 4.
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

// ---------------------------------------------------------------------------------------------------------
// k_node16s2 (round 3): the processor shape (4 + 4 k-steps, residual) with TWO 16-node tiles per
// wave and weight chunk.  One 8-wave workgroup per CU owns 16 tiles: the 320 KiB of weights cross L2 -> LDS once per CU
// instead of twice (82 instead of 164 MB per launch), every chunk barrier serves two tiles, and the two tiles' MFMA
// chains interleave inside a step.  232 VGPRs, two waves per SIMD, no scratch.
// MEASURED (TGV3D-8k x 8, bench.py timer class node_mlp): 55.2 us per launch (five steps of 64 KiB; ten steps of
// 32 KiB: 56.8) against 50.5 us for k_node16s<.., 8, 2> - SLOWER, like the 16-wave workgroup of round 2.  Neither the
// weight traffic (halved) nor the number of chunk steps (halved again) is what bounds the launch.  What does: it moves
// ~2.6 - 3 KB per node through HBM (latents + aggregates in, latents + projections out: 165 - 200 MB = 33 - 40 us at the
// 5 TB/s this access pattern gets), and since the launch is ONE round of workgroups every workgroup loads, multiplies and
// stores at the same time as all the others - memory and matrix phases do not overlap across the chip; two
// 8-tile workgroups per CU drifting out of phase (the default) recover some of that, one 16-tile workgroup does not.
// Parity green (tests/test_switches_gpu.py).  Opt-in: LB_NODE_T2=1.
template <bool PROJ>
__global__ void __launch_bounds__(512, 2)
    k_node16s2(lb_node_args a, const f32x4* __restrict__ w0h, const f32x4* __restrict__ w1h,
               const f32x4* __restrict__ wph) {
  // 64 KiB chunks (two of the 32 KiB chunks of k_node16s, which are consecutive in the packed images) through a
  // two-slot 128 KiB ring: FIVE steps - W0 rows of the latents, W0 rows of the aggregates, W1, projection halves
  constexpr int NW = 8, T = 2, CH = 2 * NS_CHUNK;
  extern __shared__ f32x4 sB2[];  // [2][CH] ring, then 192 vectors
  f32x4* const sP = sB2 + 2 * CH;
  if (a.ctrl->overflow_step >= 0) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, g = lane >> 4;
  constexpr int n_chunks = 3 + (PROJ ? 2 : 0);
  if (tid < 128) {
    const float* src = tid < 32 ? a.b0 : (tid < 64 ? a.b1 : (tid < 96 ? a.ln_s : a.ln_o));
    sP[tid] = reinterpret_cast<const f32x4*>(src)[tid & 31];
  } else if (tid < 192) {
    sP[tid] = (PROJ && a.bp) ? reinterpret_cast<const f32x4*>(a.bp)[tid - 128] : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  auto issue_chunk = [&](int c) {
    const f32x4* src = c < 2 ? w0h + (size_t)c * CH : (c == 2 ? w1h : wph + (size_t)(c - 3) * CH);
    f32x4* slot = sB2 + (c & 1) * CH;
#pragma unroll
    for (int i = 0; i < CH / 64 / NW; ++i) {
      const int piece = wave + NW * i;
      const uint32_t voff = (uint32_t)(piece * 64 + lane) * 16u;
      const uint32_t lo = (uint32_t)(uintptr_t)(lds_ptr)(slot + piece * 64);
      asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(src), "s"(lo) : "memory");
    }
  };
  // step c: chunk c has landed (it is the oldest outstanding group; everything issued later - the other slot's refill
  // and this step's own data loads - may stay in flight: vmcnt counts in order)
#define NS2_STEP(c, later_ops)                                                     \
  do {                                                                             \
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(later_ops) : "memory");               \
    if constexpr ((c) == 0) {                                                      \
      __syncthreads();                                                             \
    } else {                                                                       \
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");              \
    }                                                                              \
  } while (0)
  int64_t row[T], rowc[T];
  bool valid[T];
#pragma unroll
  for (int i = 0; i < T; ++i) {
    row[i] = (((int64_t)blockIdx.x * NW + wave) * T + i) * 16 + n;
    valid[i] = row[i] < a.n_rows;
    rowc[i] = valid[i] ? row[i] : a.n_rows - 1;
  }
  // data first (rows + CSR bounds), then the two ring slots: a wait for the rows must not cover the weights
  f32x4 va[T][8];
  int k0[T], k1[T];
#pragma unroll
  for (int i = 0; i < T; ++i) {
    const f32x4* xr = reinterpret_cast<const f32x4*>(a.xin) + rowc[i] * 32 + g;
#pragma unroll
    for (int mb = 0; mb < 8; ++mb) va[i][mb] = xr[4 * mb];
    k0[i] = k1[i] = 0;
    if (a.fused) {
      k0[i] = a.row_ptr[rowc[i]];
      k1[i] = a.row_ptr[rowc[i] + 1];
    }
  }
  issue_chunk(0);
  issue_chunk(1);
  const lds_cptr buf0 = (lds_cptr)(sB2 + lane), buf1 = (lds_cptr)(sB2 + CH + lane);
  const lds_cptr vecp = (lds_cptr)(sP + g);
  const bool probe = wave == 0;
  f32x4 acc[T][8], h0[T][4], h1[T][4];
  // ---- step 0: W0, latent rows
  NS2_STEP(0, 8);  // (chunk 1's eight pieces may still be in flight)
  if (probe) lb_range_probe(a.ctrl, va[0], 8);
#pragma unroll
  for (int i = 0; i < T; ++i) {
    lb_load_agg_half<0>(a, rowc[i], g, k0[i], k1[i], h0[i]);
    lb_load_agg_half<4>(a, rowc[i], g, k0[i], k1[i], h1[i]);
  }
#pragma unroll
  for (int i = 0; i < T; ++i) {
#pragma unroll
    for (int mb = 0; mb < 8; ++mb) acc[i][mb] = vecp[4 * mb];
    const f32x4 v01[4] = {va[i][0], va[i][1], va[i][2], va[i][3]};
    lb_gemm16v<false, 2>(buf0, v01, acc[i]);
    const f32x4 v23[4] = {va[i][4], va[i][5], va[i][6], va[i][7]};
    lb_gemm16v<false, 2>(buf0 + NS_CHUNK, v23, acc[i]);
  }
  // ---- step 1: W0, aggregated messages
  asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // chunk 1 + the aggregates
  issue_chunk(2);  // slot 0 is free: W1
  if (probe) {
    lb_range_probe(a.ctrl, h0[0], 4);
    lb_range_probe(a.ctrl, h1[0], 4);
  }
#pragma unroll
  for (int i = 0; i < T; ++i) {
    lb_gemm16v<false, 2>(buf1, h0[i], acc[i]);
    lb_gemm16v<false, 2>(buf1 + NS_CHUNK, h1[i], acc[i]);
  }
  if (probe) lb_range_probe(a.ctrl, acc[0], 8);
  // ---- step 2: W1 (ReLU folded into the operand split); the residual rows are requested behind the refill
  f32x4 acc2[T][8], res[T][8];
  asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  if constexpr (PROJ) issue_chunk(3);  // slot 1 is free
#pragma unroll
  for (int i = 0; i < T; ++i) {
    const f32x4* xr = reinterpret_cast<const f32x4*>(a.xin) + rowc[i] * 32 + g;
#pragma unroll
    for (int mb = 0; mb < 8; ++mb) res[i][mb] = xr[4 * mb];
  }
#pragma unroll
  for (int i = 0; i < T; ++i) {
#pragma unroll
    for (int mb = 0; mb < 8; ++mb) acc2[i][mb] = vecp[32 + 4 * mb];
    const f32x4 v01[4] = {acc[i][0], acc[i][1], acc[i][2], acc[i][3]};
    lb_gemm16v<true, 2>(buf0, v01, acc2[i]);
    const f32x4 v23[4] = {acc[i][4], acc[i][5], acc[i][6], acc[i][7]};
    lb_gemm16v<true, 2>(buf0 + NS_CHUNK, v23, acc2[i]);
  }
  f32x4 y[T][8];
#pragma unroll
  for (int i = 0; i < T; ++i) {
    lb_layernorm16(acc2[i], vecp + 64, vecp + 96, y[i], a.ctrl->ln_inv_d, a.ctrl->ln_pad);
#pragma unroll
    for (int mb = 0; mb < 8; ++mb) y[i][mb] = lb_pk_add(res[i][mb], y[i][mb]);
  }
  if (probe) lb_range_probe(a.ctrl, y[0], 8);
  if constexpr (PROJ) {
    // ---- steps 3, 4: projection halves [Ws | Wr]
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      if (half == 0) {
        issue_chunk(4);  // slot 0 is free
        // the node latents go out behind the refill (their drain is waited for with the next chunk, not before it)
#pragma unroll
        for (int i = 0; i < T; ++i)
          if (valid[i]) {
            f32x4* nr = reinterpret_cast<f32x4*>(a.nlat) + row[i] * 32 + g;
#pragma unroll
            for (int mb = 0; mb < 8; ++mb) nr[4 * mb] = y[i][mb];
          }
      }
      const lds_cptr bp = half == 0 ? buf1 : buf0;
#pragma unroll
      for (int i = 0; i < T; ++i) {
        f32x4 accp[8];
#pragma unroll
        for (int mb = 0; mb < 8; ++mb) accp[mb] = vecp[128 + 32 * half + 4 * mb];
        const f32x4 v01[4] = {y[i][0], y[i][1], y[i][2], y[i][3]};
        lb_gemm16v<false, 2>(bp, v01, accp);
        const f32x4 v23[4] = {y[i][4], y[i][5], y[i][6], y[i][7]};
        lb_gemm16v<false, 2>(bp + NS_CHUNK, v23, accp);
        if (valid[i]) {
          f32x4* pr = reinterpret_cast<f32x4*>(a.psr) + row[i] * 64 + 32 * half + g;
#pragma unroll
          for (int mb = 0; mb < 8; ++mb) pr[4 * mb] = accp[mb];
        }
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < T; ++i)
      if (valid[i]) {
        f32x4* nr = reinterpret_cast<f32x4*>(a.nlat) + row[i] * 32 + g;
#pragma unroll
        for (int mb = 0; mb < 8; ++mb) nr[4 * mb] = y[i][mb];
      }
  }
  (void)n_chunks;
}
#undef NS2_STEP

// ---------------------------------------------------------------------------------------------------------
// k_node16q (round 4; opt-in LB_NODE_Q=1 - built, bit-identical, NOT faster: see the launcher): the
// processor shape of k_node16s with a DEEPER weight ring in the same LDS -
// four slots of 16 KiB (one 32-wide k-step x 8 output blocks x hi|lo) instead of two of 32 KiB.  With two slots the refill
// of a slot can only be issued one chunk step (~1.3 us of MFMA time per SIMD) before it is needed - less than the
// L2 -> LDS latency under load, so every one of the ten steps waited for its weights (ablation: "weight stream 20 us" of a
// 52 us launch, nothing overlapping).  Here chunk c + 3 is requested at step c: three steps of MFMA work cover its flight.
// The in-order vmcnt makes that work only with EXACT waits: at step c the wave allows exactly the operations it has issued
// after chunk c's pieces to stay outstanding (nq_wait: later refills, residual-row loads, the stores of the node latents /
// projections - all issued unconditionally, the stores as raw-buffer stores whose bounds check drops the rows behind
// n_rows, so that every wave has the same count).  Twenty steps, one barrier each.
// Same tiles, operand order and arithmetic as k_node16s: bit-identical results.
template <bool PROJ>
__host__ __device__ constexpr int nq_ops_before_refill(int s) {  // issued by step s between its barrier and its refill
  return (s == 10 || s == 11) ? 4 : 0;   // residual rows (two halves); aggregated-message loads (steps 2, 4): a data-
                                         // dependent number, counted as 0 - they are summed as they arrive, which drains
                                         // the counter anyway (splitting request and sum costs 16 VGPRs: spills at 128)
}
template <bool PROJ>
__host__ __device__ constexpr int nq_ops_after_refill(int s) {   // issued by step s after its refill
  return s == 11 ? 8 : ((PROJ && s == 15) ? 8 : 0);  // node-latent stores; first projection half
}
template <bool PROJ>
__host__ __device__ constexpr int nq_refill(int s) { return (s >= 1 && s + 3 < (PROJ ? 20 : 12)) ? 2 : 0; }
template <bool PROJ>
__host__ __device__ constexpr int nq_wait(int c) {
  int n = 0;
  if (c < 4) {
    n = (3 - c) * 2 + 8;  // the later prologue chunks, the eight row loads
    for (int s = 0; s < c; ++s) n += nq_ops_before_refill<PROJ>(s) + nq_refill<PROJ>(s) + nq_ops_after_refill<PROJ>(s);
  } else {
    n = nq_ops_after_refill<PROJ>(c - 3);
    for (int s = c - 2; s < c; ++s) n += nq_ops_before_refill<PROJ>(s) + nq_refill<PROJ>(s) + nq_ops_after_refill<PROJ>(s);
  }
  return n;
}
#define NQ_CHUNK 1024  // f32x4 per chunk (16 KiB)
template <bool PROJ>
__global__ void __launch_bounds__(512, 4)
    k_node16q(lb_node_args a, const f32x4* __restrict__ w0h, const f32x4* __restrict__ w1h, const f32x4* __restrict__ wph) {
  __shared__ f32x4 sB[4][NQ_CHUNK];
  __shared__ f32x4 sP[192];  // [0,32) b0, [32,64) b1, [64,96) ln_s, [96,128) ln_o, [128,192) bp
  if (a.ctrl->overflow_step >= 0) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, g = lane >> 4;
  constexpr int n_chunks = PROJ ? 20 : 12;
  constexpr int C1 = 8, CP = 12;
  const float ln_inv_d = a.ctrl->ln_inv_d, ln_pad = a.ctrl->ln_pad;  // (read here: behind the chunk requests the load's wait would drain them)
  {  // bias / LayerNorm vectors -> LDS, branch-free (a load under a branch left a vmcnt(0) behind the chunk requests below:
     // the register allocator reused its destination) and complete BEFORE the first chunk is requested
    const bool has_bp = PROJ && a.bp;
    const float* src = tid < 32 ? a.b0 : (tid < 64 ? a.b1 : (tid < 96 ? a.ln_s : (tid < 128 || !has_bp ? a.ln_o : a.bp)));
    const int idx = (tid < 128 || !has_bp) ? (tid & 31) : ((tid - 128) & 63);
    f32x4 v = reinterpret_cast<const f32x4*>(src)[idx];
    if (tid >= 128 && !has_bp) v = f32x4{0.f, 0.f, 0.f, 0.f};
    if (tid < 192) sP[tid] = v;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  auto issue_chunk = [&](int c) {
    const f32x4* src = c < C1 ? w0h + (size_t)c * NQ_CHUNK : (c < CP ? w1h + (size_t)(c - C1) * NQ_CHUNK : wph + (size_t)(c - CP) * NQ_CHUNK);
    f32x4* slot = sB[c & 3];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int piece = wave + 8 * i;  // 1 KiB per wave instruction
      const uint32_t voff = (uint32_t)(piece * 64 + lane) * 16u;
      const uint32_t lo = (uint32_t)(uintptr_t)(lds_ptr)(slot + piece * 64);
      asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(src), "s"(lo) : "memory");
    }
  };
#define NQ_STEP(c)                                                            \
  do {                                                                        \
    constexpr int nw_ = nq_wait<PROJ>(c);                                     \
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(nw_) : "memory");                \
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");           \
  } while (0)
#define NQ_REFILL(c)                                                          \
  do {                                                                        \
    if constexpr ((c) >= 1 && (c) + 3 < n_chunks) issue_chunk((c) + 3);       \
  } while (0)
#define NQ_BUF(c) lane_b[(c) & 3]
  issue_chunk(0);
  issue_chunk(1);
  issue_chunk(2);
  issue_chunk(3);
  const int64_t row = ((int64_t)blockIdx.x * 8 + wave) * 16 + n;
  const int64_t rowc = row < a.n_rows ? row : a.n_rows - 1;
  lds_cptr lane_b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) lane_b[i] = (lds_cptr)(sB[i] + lane);
  const lds_cptr vecp = (lds_cptr)(sP + g);
  // stores through buffer descriptors: rows behind n_rows fall outside num_records and are dropped by the bounds check
  const __amdgpu_buffer_rsrc_t nl_rs = __builtin_amdgcn_make_buffer_rsrc(a.nlat, 0, (int)(a.n_rows * 512), 0x00020000);
  const __amdgpu_buffer_rsrc_t ps_rs = __builtin_amdgcn_make_buffer_rsrc(a.psr, 0, (int)(a.n_rows * 1024), 0x00020000);
  typedef uint32_t u32x4b __attribute__((ext_vector_type(4)));
  f32x4 va[8];
  int k0 = 0, k1 = 0;
  {
    const f32x4* xr = reinterpret_cast<const f32x4*>(a.xin) + rowc * 32 + g;
#pragma unroll
    for (int mb = 0; mb < 8; ++mb) va[mb] = xr[4 * mb];
    if (a.fused) {
      k0 = a.row_ptr[rowc];
      k1 = a.row_ptr[rowc + 1];
    }
  }
  const bool probe = wave == 0;
  f32x4 acc[8], h0[4], h1[4];
  // ---- GEMM1 over [node latents | aggregated messages]: one k-step per chunk
  NQ_STEP(0);
  if (probe) lb_range_probe(a.ctrl, va, 8);
#pragma unroll
  for (int mb = 0; mb < 8; ++mb) acc[mb] = vecp[4 * mb];
  {
    const f32x4 v[2] = {va[0], va[1]};
    lb_gemm16v<false, 1>(NQ_BUF(0), v, acc);
  }
  NQ_STEP(1);
  NQ_REFILL(1);
  {
    const f32x4 v[2] = {va[2], va[3]};
    lb_gemm16v<false, 1>(NQ_BUF(1), v, acc);
  }
  NQ_STEP(2);
  lb_load_agg_half<0>(a, rowc, g, k0, k1, h0);  // into the registers va[0..3] released
  NQ_REFILL(2);
  {
    const f32x4 v[2] = {va[4], va[5]};
    lb_gemm16v<false, 1>(NQ_BUF(2), v, acc);
  }
  NQ_STEP(3);
  NQ_REFILL(3);
  {
    const f32x4 v[2] = {va[6], va[7]};
    lb_gemm16v<false, 1>(NQ_BUF(3), v, acc);
  }
  NQ_STEP(4);
  lb_load_agg_half<4>(a, rowc, g, k0, k1, h1);
  NQ_REFILL(4);
  {
    const f32x4 v[2] = {h0[0], h0[1]};
    lb_gemm16v<false, 1>(NQ_BUF(4), v, acc);
  }
  NQ_STEP(5);
  NQ_REFILL(5);
  if (probe) lb_range_probe(a.ctrl, h0, 4);
  {
    const f32x4 v[2] = {h0[2], h0[3]};
    lb_gemm16v<false, 1>(NQ_BUF(5), v, acc);
  }
  NQ_STEP(6);
  NQ_REFILL(6);
  if (probe) lb_range_probe(a.ctrl, h1, 4);
  {
    const f32x4 v[2] = {h1[0], h1[1]};
    lb_gemm16v<false, 1>(NQ_BUF(6), v, acc);
  }
  NQ_STEP(7);
  NQ_REFILL(7);
  {
    const f32x4 v[2] = {h1[2], h1[3]};
    lb_gemm16v<false, 1>(NQ_BUF(7), v, acc);
  }
  if (probe) lb_range_probe(a.ctrl, acc, 8);
  // ---- GEMM2 (ReLU folded into the operand split)
  f32x4 acc2[8];
  NQ_STEP(8);
  NQ_REFILL(8);
#pragma unroll
  for (int mb = 0; mb < 8; ++mb) acc2[mb] = vecp[32 + 4 * mb];
  {
    const f32x4 v[2] = {acc[0], acc[1]};
    uint32_t or_h = 0;
    lb_gemm16v<true, 1, 1>(NQ_BUF(8), v, acc2, &or_h);  // per-row TINY test of the hidden row (lb_rows_tiny)
    if (lb_rows_tiny(or_h) && lane == 0) lb_raise_math(a.ctrl, LB_MATH_TINY);
  }
  NQ_STEP(9);
  NQ_REFILL(9);
  {
    const f32x4 v[2] = {acc[2], acc[3]};
    lb_gemm16v<true, 1>(NQ_BUF(9), v, acc2);
  }
  NQ_STEP(10);
  f32x4 res[8];
  const f32x4* xr = reinterpret_cast<const f32x4*>(a.xin) + rowc * 32 + g;
#pragma unroll
  for (int mb = 0; mb < 4; ++mb) res[mb] = xr[4 * mb];  // residual: second read of the node row (L2), registers of acc[0..3]
  NQ_REFILL(10);
  {
    const f32x4 v[2] = {acc[4], acc[5]};
    lb_gemm16v<true, 1>(NQ_BUF(10), v, acc2);
  }
  NQ_STEP(11);
#pragma unroll
  for (int mb = 4; mb < 8; ++mb) res[mb] = xr[4 * mb];
  {
    const f32x4 v[2] = {acc[6], acc[7]};
    lb_gemm16v<true, 1>(NQ_BUF(11), v, acc2);
  }
  // ---- LayerNorm + residual
  f32x4 y[8];
  lb_layernorm16(acc2, vecp + 64, vecp + 96, y, ln_inv_d, ln_pad);
#pragma unroll
  for (int mb = 0; mb < 8; ++mb) y[mb] = lb_pk_add(res[mb], y[mb]);
  // this step's refill goes out HERE: the compiler's wait for the residual rows above does not know the chunk requests and
  // would drain one issued before it (same position in the operation order as far as nq_wait is concerned: rows, refill, stores)
  NQ_REFILL(11);
  if (probe) lb_range_probe(a.ctrl, y, 8);
  {
    const uint32_t off = (uint32_t)row * 512u + (uint32_t)g * 16u;
#pragma unroll
    for (int mb = 0; mb < 8; ++mb) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4b, y[mb]), nl_rs, (int)off, 64 * mb, 0);
  }
  // ---- projection for the next edge MLP: psr = y @ [Ws | Wr] + [0 | b0_next], one 128-wide half at a time
  if constexpr (PROJ) {
#define NQ_PROJ_STEP(c, i0)                                      \
  NQ_STEP(c);                                                    \
  NQ_REFILL(c);                                                  \
  {                                                              \
    const f32x4 v[2] = {y[i0], y[i0 + 1]};                       \
    lb_gemm16v<false, 1>(NQ_BUF(c), v, accp);                    \
  }
    {
      f32x4 accp[8];
#pragma unroll
      for (int mb = 0; mb < 8; ++mb) accp[mb] = vecp[128 + 4 * mb];
      NQ_PROJ_STEP(12, 0)
      NQ_PROJ_STEP(13, 2)
      NQ_PROJ_STEP(14, 4)
      NQ_PROJ_STEP(15, 6)
      const uint32_t off = (uint32_t)row * 1024u + (uint32_t)g * 16u;
#pragma unroll
      for (int mb = 0; mb < 8; ++mb)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4b, accp[mb]), ps_rs, (int)off, 64 * mb, 0);
    }
    {
      f32x4 accp[8];
#pragma unroll
      for (int mb = 0; mb < 8; ++mb) accp[mb] = vecp[128 + 32 + 4 * mb];
      NQ_PROJ_STEP(16, 0)
      NQ_PROJ_STEP(17, 2)
      NQ_PROJ_STEP(18, 4)
      NQ_PROJ_STEP(19, 6)
      const uint32_t off = (uint32_t)row * 1024u + 512u + (uint32_t)g * 16u;
#pragma unroll
      for (int mb = 0; mb < 8; ++mb)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4b, accp[mb]), ps_rs, (int)off, 64 * mb, 0);
    }
#undef NQ_PROJ_STEP
  }
#undef NQ_STEP
#undef NQ_REFILL
#undef NQ_BUF
}

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
  // LB_NODE_T2=1: two tiles per wave and weight chunk (k_node16s2) for the processor shape on batches
  static const int node_q = getenv("LB_NODE_Q") ? atoi(getenv("LB_NODE_Q")) : 0;  // 1: batches, 2: also small launches (tests)
  static const int t2 = getenv("LB_NODE_T2") ? atoi(getenv("LB_NODE_T2")) : 0;  // 2: also on small launches (tests)
  if (t2 && npa == 4 && npb == 4 && resid && (nw == 8 || t2 == 2)) {
    dim3 grid2((unsigned)((tiles + 15) / 16)), block2(512);
    const size_t lds2 = sizeof(f32x4) * (4 * NS_CHUNK + 192);  // 128 KiB ring + the bias / LayerNorm vectors
    static bool raised = false;
    if (!raised) {
      (void)hipFuncSetAttribute((const void*)k_node16s2<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
      (void)hipFuncSetAttribute((const void*)k_node16s2<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
      raised = true;
    }
    if (e->ext_armed && !e->ext_used) {
      if (proj)
        hipExtLaunchKernelGGL((k_node16s2<true>), grid2, block2, lds2, e->stream, e->trecs.back().a, e->trecs.back().b, 0, a, w0, w1, wp);
      else
        hipExtLaunchKernelGGL((k_node16s2<false>), grid2, block2, lds2, e->stream, e->trecs.back().a, e->trecs.back().b, 0, a, w0, w1, wp);
      e->ext_used = true;
    } else if (proj) {
      hipLaunchKernelGGL((k_node16s2<true>), grid2, block2, lds2, e->stream, a, w0, w1, wp);
    } else {
      hipLaunchKernelGGL((k_node16s2<false>), grid2, block2, lds2, e->stream, a, w0, w1, wp);
    }
  } else if (npa == 4 && npb == 4 && resid && proj && ((node_q && nw == 8) || node_q == 2) && a.n_rows < (1 << 21)) {
    // round 4, opt-in (LB_NODE_Q=1): four-slot ring of 16 KiB chunks (k_node16q) - parity green, measured 1.5 % SLOWER than
    // the two-slot kernel (TGV3D x 8: 51.4 vs 50.6 us per launch, profiles/r04_node_q_ab.txt); the last layer (no projection)
    // stays on k_node16s
    dim3 gridq((unsigned)((tiles + 7) / 8));
    LB_LAUNCH_TIMED(e, (k_node16q<true>), gridq, dim3(512), a, w0, w1, wp);
  } else if (npa == 4 && npb == 4 && resid)
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
