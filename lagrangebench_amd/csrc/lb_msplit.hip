// lb_msplit.hip - "M-split" network kernels for SMALL graphs (one trajectory per GPU: the BASELINE configs as
// literally stated), round 3.
//
// Reference functions replaced: GNS._encoder / _processor (edge + node update, jraph.segment_sum, residuals) of
// lagrangebench/models/gns.py:65-124, build_mlp (hk.nets.MLP + hk.LayerNorm) of models/utils.py:100-115.
//
// Why a second kernel family.  The round-2 kernels give every WAVE a whole 16-row tile and all 128 output features,
// so a wave needs the complete weight matrices: 128 KiB (edge MLP) are staged global -> registers -> LDS by every
// workgroup of every launch, 320 KiB (node MLP + next projection) are streamed through an LDS ring in ten
// barrier-separated chunks.  On a 64 k-node batch that staging is amortised over dozens of tiles per wave; on ONE
// 2.5 k - 8 k particle trajectory a wave has ~1 tile, a launch is the latency chain "stage weights -> one tile" and
// the step is 30 such chains (round 2: 13 + 15 us per layer on TGV2D-2.5k, of which ~2 us is arithmetic).
// Here the 4 waves of a workgroup split the OUTPUT features (M) of every Linear instead of the rows:
//   * wave w owns output blocks 2w, 2w+1 (32 features) of each 128-wide Linear (four blocks of the 256-wide
//     projection), so it needs 1/4 of every matrix - 32 KiB of an edge MLP, 80 KiB of node MLP + projection - which
//     it loads STRAIGHT INTO REGISTERS as MFMA A-operand fragments (128 / 320 VGPRs): no LDS staging, no chunk
//     barriers, the loads are in flight while the tile's indices / gathers are;
//   * the B operand (a 16-row tile of activations, fp16 hi | lo) is exchanged through LDS in MFMA fragment order:
//     8 KiB per tile and Linear instead of 128 - 320 KiB of weights; the register-chaining trick of the round-2
//     kernels survives as a K-permutation folded into the packed weights (lane (n, g) of wave w holds features
//     32w + 16c + 4g + j in the C layout and writes them as elements 4c + j of k-group g of k-block w);
//   * LayerNorm statistics are combined across the 4 waves with Chan's parallel (mean, M2) update: one exchange;
//   * cross-lane reductions use DPP rotations and gfx950's v_permlane16/32_swap - no ds_bpermute round trips;
//   * two independent workgroups share a CU (edge kernels: <= 256 VGPRs), so one's barrier / LDS / memory latency
//     is the other's issue slot.
// Arithmetic = the f16x2 scheme of lb_f16x2.h (lo*hi + hi*lo + hi*hi on v_mfma_f32_16x16x32_f16, fp32 accumulate).
// The f16x2 range guard is EXHAUSTIVE here: every GEMM operand of every tile is folded into a running maximum
// (LARGE / non-finite) and a per-tile maximum combined across the waves (TINY), one atomicOr per wave and launch.
#include <stdio.h>
#include <stdlib.h>
#include <stddef.h>
#include <string.h>

#include <algorithm>

#include "lb_msplit_dev.h"
#include "lb_features.h"

typedef _Float16 h4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------ host packing
static uint16_t ms_f32_to_f16(float f) {
  const _Float16 h = (_Float16)f;  // RNE
  uint16_t u;
  memcpy(&u, &h, 2);
  return u;
}
static float ms_f16_to_f32(uint16_t u) {
  _Float16 h;
  memcpy(&h, &u, 2);
  return (float)h;
}

// W is (K, M) row-major (haiku Linear).  out[((mb*nkb + kb)*2 + part)*64 + lane] (16 B = 8 halfs each): the A
// fragment of output block mb (features 16 mb ..) and k-block kb; part 0 = hi, 1 = lo.  (wv, npw only enumerate the
// blocks: mb = wv*npw + c - the image does not depend on how the waves split them.)
// perm: k-block kb, k-group kg, element i <-> input feature 32 kb + 16 (i >> 2) + 4 kg + (i & 3) (the C-layout
// exchange above); natural: 32 kb + 8 kg + i.
void lb_pack_ms(const float* w, int K, int M, int nkb, int npw, bool perm, float* out) {
  uint16_t* o = reinterpret_cast<uint16_t*>(out);
  for (int wv = 0; wv < 8; ++wv)
    for (int c = 0; c < npw; ++c)
      for (int kb = 0; kb < nkb; ++kb)
        for (int lane = 0; lane < 64; ++lane)
          for (int i = 0; i < 8; ++i) {
            const int kg = lane >> 4;
            const int m = 16 * (wv * npw + c) + (lane & 15);
            const int k = perm ? 32 * kb + 16 * (i >> 2) + 4 * kg + (i & 3) : 32 * kb + 8 * kg + i;
            const float x = (k < K && m < M) ? w[(size_t)k * M + m] : 0.f;
            const uint16_t hi = ms_f32_to_f16(x);
            const uint16_t lo = ms_f32_to_f16(x - ms_f16_to_f32(hi));
            const size_t f = (((size_t)(wv * npw + c) * nkb + kb) * 2) * 64;
            o[(f + lane) * 8 + i] = hi;
            o[(f + 64 + lane) * 8 + i] = lo;
          }
}

// =========================================================================================== edge kernels
// Processor edge MLP + residual + fused segment_sum (gns.py:86-101,117-122), one tile of 16 edges per iteration.
// Per tile and wave: 2 KiB of latents (blocks 2w, 2w+1 of the tile-blocked layout = exactly the C layout of this
// wave's output blocks: also the residual operand), four 16-B gathers (sender / receiver projection slices) as
// accumulator start, 24 + 24 MFMAs, an 8-register segmented scan.  The next tile's loads are issued before this
// tile's GEMMs and taken delivery of before its stores (gfx9's single in-order vmcnt: a wait behind the stores
// would be a store drain).
template <bool SKIP>
__global__ void __launch_bounds__(MS_THREADS, 2) k_edge_ms(lb_ems_args a) {
  __shared__ f32x4 sB1[4 * 2 * 64];
  __shared__ f32x4 sB2[4 * 2 * 64];
  __shared__ __attribute__((aligned(16))) f32x2m sRed[16 * 4];
  __shared__ __attribute__((aligned(16))) int sMx[2][2][MS_GUARD_WORDS];
  const int poisoned = a.ctrl->overflow_step;
  const int E = a.ctrl->n_edges_total;
  const float ln_inv_d = a.ctrl->ln_inv_d, ln_pad = a.ctrl->ln_pad;
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, g = lane >> 4;
  MS_STAMP(0);
  // XCD-contiguous tile walk: the receivers of an XCD's edge tiles are (mostly) the nodes of the same XCD's node
  // tiles, so the aggregated messages / projections cross kernels through that XCD's L2.  (Tried in round 3: a
  // round-robin walk whose first tile does not wait for the edge count - the edge launch gains 0.1 us, the node
  // launch that reads agg / part from seven other L2s loses 1 us.)
  const int ntiles = (E + 15) >> 4;
  ms_walk wk;
  if (poisoned >= 0 || !wk.init(ntiles)) return;  // uniform over the workgroup
  const f32x4* psr4 = reinterpret_cast<const f32x4*>(a.psr);
  const f32x4* elat4 = reinterpret_cast<const f32x4*>(a.elat);

  f32x4 nve[2], nps[2], npr[2];
  int nr, nrb, s_i, r_i;
  auto load_idx = [&](int t) {
    const int row = t * 16 + n;
    const int rc = row < E ? row : E - 1;
    s_i = a.senders[rc];
    r_i = a.receivers[rc];
  };
  auto issue = [&](int t) {
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      nve[c] = elat4[((int64_t)t * 8 + 2 * w + c) * 64 + lane];
      nps[c] = psr4[(int64_t)s_i * 64 + 8 * w + 4 * c + g];
      npr[c] = psr4[(int64_t)r_i * 64 + 32 + 8 * w + 4 * c + g];
    }
    nr = r_i;
    nrb = lb_edge_probe(a.receivers, t, lane, E);
  };
  // Launch prologue = the latency chain "control block -> indices -> gathers": the first tile's loads go out
  // BEFORE the weights (vmcnt retires in order: a wait for the gathers must not cover the 32 KiB of weights)
  int t = wk.q;
  load_idx(t);
  issue(t);
  load_idx(min(t + wk.stride, wk.q_last));
  // this wave's weight fragments, straight into registers
  h8 w0h[2][4], w0l[2][4], w1h[2][4], w1l[2][4];
  {
    const f32x4* wb = reinterpret_cast<const f32x4*>(a.w) + lane;
    ms_wload<4, 2>(wb, 2 * w, w0h, w0l);
    ms_wload<4, 2>(wb + 8 * 4 * 2 * 64, 2 * w, w1h, w1l);
  }
  f32x4 b1v[2], lns[2], lno[2];
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    b1v[c] = reinterpret_cast<const f32x4*>(a.b1)[8 * w + 4 * c + g];
    lns[c] = reinterpret_cast<const f32x4*>(a.ln_s)[8 * w + 4 * c + g];
    lno[c] = reinterpret_cast<const f32x4*>(a.ln_o)[8 * w + 4 * c + g];
  }
  ms_guard guard{0.f, 0};

  // (range-guard slots: OR-accumulated per tile, cleared one iteration ahead - see ms_guard)
  for (int i = tid; i < (int)(sizeof(sMx) / sizeof(int)); i += MS_THREADS) reinterpret_cast<int*>(sMx)[i] = 0;
  __syncthreads();
  for (int it = 0; it < wk.n_iter; ++it, t += wk.stride) {
    f32x4 ve[2], acc[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      ve[c] = nve[c];
      acc[c] = nps[c] + npr[c];
    }
    const int rcur = nr, rb = nrb;
    if (it < 3) MS_STAMP(1 + 8 * it);
    uint32_t orv;
    ms_stage<false>(sB1, w, lane, ve[0], ve[1], orv, guard.big);
    guard.post(sMx[it & 1][0], orv);
    // the next tile's loads (and the indices of the one after) go out before this tile's GEMMs
    issue(min(t + wk.stride, wk.q_last));
    load_idx(min(t + 2 * wk.stride, wk.q_last));
    if (it < 3) MS_STAMP(2 + 8 * it);
    __syncthreads();
    if (it < 3) MS_STAMP(3 + 8 * it);
    ms_gemm<4, 2>(sB1, lane, w0h, w0l, acc);
    guard.tile_codes(sMx[it & 1][0], sMx[(it & 1) ^ 1][0]);
    uint32_t orv2;
    ms_stage<true>(sB2, w, lane, acc[0], acc[1], orv2, guard.big);
    guard.post(sMx[it & 1][1], orv2);
    if (it < 3) MS_STAMP(4 + 8 * it);
    __syncthreads();
    if (it < 3) MS_STAMP(5 + 8 * it);
    f32x4 acc2[2] = {b1v[0], b1v[1]};
    ms_gemm<4, 2>(sB2, lane, w1h, w1l, acc2);
    guard.tile_codes(sMx[it & 1][1], sMx[(it & 1) ^ 1][1]);
    {
      const f32x2m p = ms_ln_local(acc2[0], acc2[1]);
      if (g == 0) sRed[n * 4 + w] = p;
    }
    if (it < 3) MS_STAMP(6 + 8 * it);
    __syncthreads();
    if (it < 3) MS_STAMP(7 + 8 * it);
    float mean, rs;
    ms_ln_combine(sRed, n, ln_inv_d, ln_pad, mean, rs);
    f32x4 y[2];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int j = 0; j < 4; ++j) y[c][j] = (lns[c][j] * rs) * (acc2[c][j] - mean) + lno[c][j];
    // take delivery of the prefetched tile HERE, while only loads are in flight
    asm volatile("" : "+v"(nve[0]), "+v"(nve[1]), "+v"(nps[0]), "+v"(nps[1]), "+v"(npr[0]), "+v"(npr[1]), "+v"(nrb));
    if (it < 3) MS_STAMP(8 + 8 * it);
    const int row = t * 16 + n;
    const bool valid = row < E;
    if (!SKIP) {
      f32x4* ew = reinterpret_cast<f32x4*>(a.elat) + ((int64_t)t * 8 + 2 * w) * 64 + lane;
      ew[0] = ve[0] + y[0];  // residual (gns.py:120-122)
      ew[64] = ve[1] + y[1];
    }
    // fused jraph.segment_sum: segmented Hillis-Steele scan over the 16 edges of the DPP row
    const int rr = valid ? rcur : (-1 - n);
    const int r_prev = __builtin_amdgcn_update_dpp(-2, rr, 0x111, 0xF, 0xF, false);
    const bool head = (n == 0) || (rr != r_prev);
    const unsigned H = (unsigned)(__ballot(head) & 0xffffull);
    const unsigned below = H & ((2u << n) - 1u);
    const int segstart = 31 - __clz(below);
    const bool tail = (n == 15) || ((H >> (n + 1)) & 1u);
    const float m1 = (n >= 1 && segstart <= n - 1) ? 1.f : 0.f, m2 = (n >= 2 && segstart <= n - 2) ? 1.f : 0.f;
    const float m4 = (n >= 4 && segstart <= n - 4) ? 1.f : 0.f, m8 = (n >= 8 && segstart <= n - 8) ? 1.f : 0.f;
    if (!valid) {
      y[0] = f32x4{0.f, 0.f, 0.f, 0.f};
      y[1] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    lb_scan8(y[0], y[1], m1, m2, m4, m8);
    if (tail && valid) {
      int slot01;
      const bool complete = lb_seg_complete(rb, rr, segstart, n, t, E, slot01);
      float* dst = complete ? a.agg + (int64_t)rr * 128 : a.part + ((int64_t)t * 2 + slot01) * 128;
      f32x4* d4 = reinterpret_cast<f32x4*>(dst) + 8 * w + g;
      d4[0] = y[0];
      d4[4] = y[1];
    }
  }
  MS_STAMP(30);
  guard.commit(a.ctrl, lane);
  MS_STAMP(31);
}

// Encoder edge MLP (gns.py:73-84): e0 = LayerNorm(W1 relu(W0 f + b0) + b1), f = (rel_disp, rel_dist) zero-padded
// to 8 floats.  The first Linear is ONE k-block whose B operand every wave builds itself from the 32-B feature row
// (natural k order: only k-group 0 is non-zero) - no exchange; then as above without gathers / residual / scan.
__global__ void __launch_bounds__(MS_THREADS, 2) k_edge_enc_ms(lb_ems_args a) {
  __shared__ f32x4 sB2[4 * 2 * 64];
  __shared__ __attribute__((aligned(16))) f32x2m sRed[16 * 4];
  __shared__ __attribute__((aligned(16))) int sMx[2][1][MS_GUARD_WORDS];
  const int poisoned = a.ctrl->overflow_step;
  const int E = a.ctrl->n_edges_total;
  const float ln_inv_d = a.ctrl->ln_inv_d, ln_pad = a.ctrl->ln_pad;
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, g = lane >> 4;
  h8 w0h[2][1], w0l[2][1], w1h[2][4], w1l[2][4];
  {
    const f32x4* wb = reinterpret_cast<const f32x4*>(a.w) + lane;
    ms_wload<1, 2>(wb, 2 * w, w0h, w0l);
    ms_wload<4, 2>(wb + 8 * 1 * 2 * 64, 2 * w, w1h, w1l);
  }
  f32x4 b0v[2], b1v[2], lns[2], lno[2];
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    b0v[c] = reinterpret_cast<const f32x4*>(a.b0)[8 * w + 4 * c + g];
    b1v[c] = reinterpret_cast<const f32x4*>(a.b1)[8 * w + 4 * c + g];
    lns[c] = reinterpret_cast<const f32x4*>(a.ln_s)[8 * w + 4 * c + g];
    lno[c] = reinterpret_cast<const f32x4*>(a.ln_o)[8 * w + 4 * c + g];
  }
  const int ntiles = (E + 15) >> 4;
  ms_walk wk;
  if (poisoned >= 0 || !wk.init(ntiles)) return;
  const f32x4* ef4 = reinterpret_cast<const f32x4*>(a.efeat);
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  f32x4 fn[2];
  auto issue = [&](int t) {
    const int row = t * 16 + n;
    const int64_t rc = row < E ? row : E - 1;
    fn[0] = ef4[rc * 2];
    fn[1] = ef4[rc * 2 + 1];
  };
  int t = wk.q;
  issue(t);
  ms_guard guard{0.f, 0};
  // (range-guard slots: OR-accumulated per tile, cleared one iteration ahead - see ms_guard)
  for (int i = tid; i < (int)(sizeof(sMx) / sizeof(int)); i += MS_THREADS) reinterpret_cast<int*>(sMx)[i] = 0;
  __syncthreads();
  for (int it = 0; it < wk.n_iter; ++it, t += wk.stride) {
    f32x4 acc[2] = {b0v[0], b0v[1]};
    {
      h8 bh, bl;
      lb_split8v(g == 0 ? fn[0] : zero, g == 0 ? fn[1] : zero, bh, bl);
#pragma unroll
      for (int c = 0; c < 2; ++c) acc[c] = MFMA16H(w0l[c][0], bh, acc[c]);
#pragma unroll
      for (int c = 0; c < 2; ++c) acc[c] = MFMA16H(w0h[c][0], bl, acc[c]);
#pragma unroll
      for (int c = 0; c < 2; ++c) acc[c] = MFMA16H(w0h[c][0], bh, acc[c]);
    }
    issue(min(t + wk.stride, wk.q_last));
    uint32_t orv;
    ms_stage<true>(sB2, w, lane, acc[0], acc[1], orv, guard.big);
    guard.post(sMx[it & 1][0], orv);
    __syncthreads();
    f32x4 acc2[2] = {b1v[0], b1v[1]};
    ms_gemm<4, 2>(sB2, lane, w1h, w1l, acc2);
    guard.tile_codes(sMx[it & 1][0], sMx[(it & 1) ^ 1][0]);
    {
      const f32x2m p = ms_ln_local(acc2[0], acc2[1]);
      if (g == 0) sRed[n * 4 + w] = p;
    }
    __syncthreads();
    float mean, rs;
    ms_ln_combine(sRed, n, ln_inv_d, ln_pad, mean, rs);
    asm volatile("" : "+v"(fn[0]), "+v"(fn[1]));
    f32x4* ew = reinterpret_cast<f32x4*>(a.elat) + ((int64_t)t * 8 + 2 * w) * 64 + lane;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      f32x4 y;
#pragma unroll
      for (int j = 0; j < 4; ++j) y[j] = (lns[c][j] * rs) * (acc2[c][j] - mean) + lno[c][j];
      ew[64 * c] = y;
    }
    // (sB2 / sRed / sMx are next written after barriers every wave reaches only once it is done reading them)
  }
  guard.commit(a.ctrl, lane);
}

// =========================================================================================== node kernel
// Node MLP (+ residual + projection for the next edge MLP): encoder node branch (gns.py:65-72) with NKA k-blocks of
// input features, or processor update_node_features (gns.py:103-113,120-122) with NKA = 4 latents + 4 k-blocks of
// aggregated messages.  Wave w owns output blocks 2w, 2w+1 of both Linears and 4w .. 4w+3 of the 256-wide
// projection: 320 registers of weights (one wave per SIMD, 512 registers each).  T tiles of 16 nodes per iteration
// (their loads are in flight together; the weights are reused from registers).
// DEC (last layer, no projection): decoder MLP + (rollout step) integrator in the epilogue - one launch less per step.
// Round 5: the body is a device function over caller-provided LDS (ms_node_smem) so that the fused layer kernel of
// lb_own.hip - edge phase + node phase of one message-passing layer in ONE launch - can run it behind its edge phase
// in the LDS the edge weights occupied.  OWN (that kernel): the partial-sum slots are indexed by OWNER tiles - the
// 16-edge tiles of a node tile's own edge range, slot (row_ptr[16 q] >> 4) + q + j (lb_own.hip) - instead of the global
// 16-edge tiles.
template <int NKA, bool AGG, int T, bool PD>
struct ms_node_smem {
  static constexpr int NK0 = NKA + (AGG ? 4 : 0);
  f32x4 (*sB1)[NK0 * 2 * 64];
  f32x4 (*sB2)[4 * 2 * 64];
  f32x4 (*sB3)[4 * 2 * 64];
  f32x2m (*sRed)[16 * 4];
  int (*sMx)[4 * T][MS_GUARD_WORDS];
  static constexpr int f32x4_count = T * NK0 * 128 + T * 512 + (PD ? T : 1) * 512 + T * 32 + 2 * 4 * T * MS_GUARD_WORDS / 4;
  __device__ __forceinline__ void carve(f32x4* base) {
    sB1 = reinterpret_cast<f32x4(*)[NK0 * 2 * 64]>(base);
    base += T * NK0 * 128;
    sB2 = reinterpret_cast<f32x4(*)[4 * 2 * 64]>(base);
    base += T * 512;
    sB3 = reinterpret_cast<f32x4(*)[4 * 2 * 64]>(base);
    base += (PD ? T : 1) * 512;
    sRed = reinterpret_cast<f32x2m(*)[16 * 4]>(base);
    base += T * 32;
    sMx = reinterpret_cast<int(*)[4 * T][MS_GUARD_WORDS]>(base);
  }
};

template <int NKA, bool AGG, bool RESID, bool PROJ, int T, bool DEC, bool OWN = false>
__device__ __forceinline__ void ms_node_body(const lb_nms_args& a, const lb_geom& geom,
                                             const ms_node_smem<NKA, AGG, T, (PROJ || DEC)>& sm) {
  static_assert(!(DEC && PROJ), "the decoder follows the LAST layer");
  constexpr int NK0 = NKA + (AGG ? 4 : 0);
  auto& sB1 = sm.sB1;
  auto& sB2 = sm.sB2;
  auto& sB3 = sm.sB3;
  auto& sRed = sm.sRed;
  auto& sMx = sm.sMx;
  const int poisoned = a.ctrl->overflow_step;
  const int step = a.ctrl->step;
  const float ln_inv_d = a.ctrl->ln_inv_d, ln_pad = a.ctrl->ln_pad;
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, g = lane >> 4;
  const int ntiles = (int)((a.n_rows + 15) >> 4);
  ms_walk wk;
  // (acting on the poison flag only after the first loads are out - the trick of the edge kernels - measured SLOWER
  // here: 9.4 -> 10.3 us per launch on TGV2D-2.5k)
  const bool has_work = wk.init((ntiles + T - 1) / T);
  if (poisoned >= 0) return;
  if (!has_work) {
    if constexpr (DEC) {  // (an idle workgroup still counts towards "everybody has finished")
      if (a.integ.on && tid == 0 && atomicAdd(a.integ.blocks_done, 1) == (int)gridDim.x - 1) {
        *a.integ.blocks_done = 0;
        const_cast<lb_ctrl*>(a.ctrl)->step = step + 1;
      }
    }
    return;
  }
  const f32x4* xin4 = reinterpret_cast<const f32x4*>(a.xin);
  const bool has_x = w < NKA;  // this wave holds an input k-block (uniform)
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};

  // one iteration's inputs: rows + (fused aggregation) the per-row sources.  Issued BEFORE the weights on the
  // first iteration (vmcnt retires in order: the wait for the rows must not cover 80 KiB of weights).
  f32x4 xa[T][2], ag[T][2];
  bool valid[T];
  int64_t rcv[T];
  auto load_rows = [&](int q) {
    int k0[T], k1[T];
#pragma unroll
    for (int i = 0; i < T; ++i) {
      const int64_t row = ((int64_t)q * T + i) * 16 + n;
      valid[i] = row < a.n_rows;
      rcv[i] = valid[i] ? row : a.n_rows - 1;
      xa[i][0] = xa[i][1] = zero;
      if (has_x) {
        xa[i][0] = xin4[rcv[i] * (8 * NKA) + 8 * w + g];
        xa[i][1] = xin4[rcv[i] * (8 * NKA) + 8 * w + 4 + g];
      }
      if constexpr (AGG) {
        if (a.fused) {
          k0[i] = a.row_ptr[rcv[i]];
          k1[i] = a.row_ptr[rcv[i] + 1];
        }
      }
    }
    if constexpr (AGG) {
#pragma unroll
      for (int i = 0; i < T; ++i) {
        const int64_t rc = rcv[i];
        if (!a.fused) {
          ag[i][0] = reinterpret_cast<const f32x4*>(a.agg)[rc * 32 + 8 * w + g];
          ag[i][1] = reinterpret_cast<const f32x4*>(a.agg)[rc * 32 + 8 * w + 4 + g];
        } else {
          // one source when the receiver's CSR row lies inside one 16-edge tile (agg[r]), else the per-tile
          // partial slots in tile order (epilogues of k_edge_ms / k_edge16v)
          // OWN: tiles counted from the node tile's first edge R0, slots shifted by (R0 >> 4) + node tile
          int r0 = 0, sbase = 0;
          if constexpr (OWN) {
            const int64_t qt = (int64_t)q * T + i;
            r0 = a.row_ptr[qt * 16];
            sbase = (r0 >> 4) + (int)qt;
          }
          const int t0 = (k0[i] - r0) >> 4, t1 = (k1[i] - 1 - r0) >> 4;
          const bool single = t0 == t1;
          const int nsrc = (k1[i] <= k0[i]) ? 0 : (single ? 1 : t1 - t0 + 1);
          const int kk0 = k0[i];
          auto slot_of = [&](int tt) -> const f32x4* {
            const float* src = single ? a.agg + rc * 128
                                      : a.part + ((int64_t)(sbase + tt) * 2 + (kk0 <= r0 + (tt << 4) ? 0 : 1)) * 128;
            return reinterpret_cast<const f32x4*>(src) + 8 * w + g;
          };
          // the first two sources together (a row of ~7-17 edges usually straddles at most one tile boundary)
          const f32x4* s0 = slot_of(t0);
          const f32x4* s1 = nsrc >= 2 ? slot_of(t0 + 1) : s0;
          const f32x4 v00 = s0[0], v01 = s0[4], v10 = s1[0], v11 = s1[4];
          ag[i][0] = (nsrc >= 1 ? v00 : zero) + (nsrc >= 2 ? v10 : zero);
          ag[i][1] = (nsrc >= 1 ? v01 : zero) + (nsrc >= 2 ? v11 : zero);
          for (int s = 2; __any(s < nsrc); ++s)
            if (s < nsrc) {
              const f32x4* sp = slot_of(t0 + s);
              ag[i][0] = ag[i][0] + sp[0];
              ag[i][1] = ag[i][1] + sp[4];
            }
        }
      }
    }
  };
  MS_STAMP(0);
  int q = wk.q;
  load_rows(q);
  MS_STAMP(1);

  h8 w0h[2][NK0], w0l[2][NK0], w1h[2][4], w1l[2][4], wph[PROJ ? 4 : 1][4], wpl[PROJ ? 4 : 1][4];
  {
    const f32x4* wb = reinterpret_cast<const f32x4*>(a.w) + lane;
    ms_wload<NK0, 2>(wb, 2 * w, w0h, w0l);
    const f32x4* wb1 = wb + 8 * NK0 * 2 * 64;
    ms_wload<4, 2>(wb1, 2 * w, w1h, w1l);
    if constexpr (PROJ) ms_wload<4, 4>(wb1 + 8 * 4 * 2 * 64, 4 * w, wph, wpl);
  }
  h8 wd0h[DEC ? 2 : 1][4], wd0l[DEC ? 2 : 1][4], wd1h[1][4], wd1l[1][4];
  f32x4 bd0v[2] = {};
  if constexpr (DEC) {
    const f32x4* wbd = reinterpret_cast<const f32x4*>(a.w) + lane + 8 * NK0 * 2 * 64 + 8 * 4 * 2 * 64;
    ms_wload<4, 2>(wbd, 2 * w, wd0h, wd0l);
    ms_wload<4, 1>(wbd + 8 * 4 * 2 * 64, 0, wd1h, wd1l);
    bd0v[0] = reinterpret_cast<const f32x4*>(a.bd0)[8 * w + g];
    bd0v[1] = reinterpret_cast<const f32x4*>(a.bd0)[8 * w + 4 + g];
  }
  f32x4 b0v[2], b1v[2], lns[2], lno[2], bpv[4];
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    b0v[c] = reinterpret_cast<const f32x4*>(a.b0)[8 * w + 4 * c + g];
    b1v[c] = reinterpret_cast<const f32x4*>(a.b1)[8 * w + 4 * c + g];
    lns[c] = reinterpret_cast<const f32x4*>(a.ln_s)[8 * w + 4 * c + g];
    lno[c] = reinterpret_cast<const f32x4*>(a.ln_o)[8 * w + 4 * c + g];
  }
#pragma unroll
  for (int c = 0; c < 4; ++c)
    bpv[c] = PROJ ? reinterpret_cast<const f32x4*>(a.bp)[16 * w + 4 * c + g] : zero;
  ms_guard guard{0.f, 0};

  MS_STAMP(2);
  // (range-guard slots: OR-accumulated per tile, cleared one iteration ahead - see ms_guard)
  for (int i = tid; i < 2 * 4 * T * MS_GUARD_WORDS; i += MS_THREADS) reinterpret_cast<int*>(sMx)[i] = 0;
  __syncthreads();
  for (int it = 0; it < wk.n_iter; ++it, q += wk.stride) {
    if (it > 0) load_rows(q);
    if (it < 2) MS_STAMP(3 + 12 * it);
#pragma unroll
    for (int i = 0; i < T; ++i) {
      uint32_t orv = 0u, orv2 = 0u;  // the tile's first operand is [rows | aggregated messages]: one code for both
      if (has_x) ms_stage<false>(sB1[i], w, lane, xa[i][0], xa[i][1], orv, guard.big);
      if constexpr (AGG) ms_stage<false>(sB1[i], NKA + w, lane, ag[i][0], ag[i][1], orv2, guard.big);
      guard.post(sMx[it & 1][(0) * T + (i)], orv | orv2);
    }
    if (it < 2) MS_STAMP(4 + 12 * it);
    __syncthreads();
    if (it < 2) MS_STAMP(5 + 12 * it);
    f32x4 acc[T][2];
#pragma unroll
    for (int i = 0; i < T; ++i) {
      acc[i][0] = b0v[0];
      acc[i][1] = b0v[1];
      ms_gemm<NK0, 2>(sB1[i], lane, w0h, w0l, acc[i]);
    }
#pragma unroll
    for (int i = 0; i < T; ++i) {
      guard.tile_codes(sMx[it & 1][(0) * T + (i)], sMx[(it & 1) ^ 1][(0) * T + (i)]);
      uint32_t orv;
      ms_stage<true>(sB2[i], w, lane, acc[i][0], acc[i][1], orv, guard.big);
      guard.post(sMx[it & 1][(1) * T + (i)], orv);
    }
    if (it < 2) MS_STAMP(6 + 12 * it);
    __syncthreads();
    if (it < 2) MS_STAMP(7 + 12 * it);
    f32x4 acc2[T][2];
#pragma unroll
    for (int i = 0; i < T; ++i) {
      acc2[i][0] = b1v[0];
      acc2[i][1] = b1v[1];
      ms_gemm<4, 2>(sB2[i], lane, w1h, w1l, acc2[i]);
    }
#pragma unroll
    for (int i = 0; i < T; ++i) {
      guard.tile_codes(sMx[it & 1][(1) * T + (i)], sMx[(it & 1) ^ 1][(1) * T + (i)]);
      const f32x2m p = ms_ln_local(acc2[i][0], acc2[i][1]);
      if (g == 0) sRed[i][n * 4 + w] = p;
    }
    if (it < 2) MS_STAMP(8 + 12 * it);
    __syncthreads();
    if (it < 2) MS_STAMP(9 + 12 * it);
    f32x4 y[T][2];
#pragma unroll
    for (int i = 0; i < T; ++i) {
      float mean, rs;
      ms_ln_combine(sRed[i], n, ln_inv_d, ln_pad, mean, rs);
#pragma unroll
      for (int c = 0; c < 2; ++c) {
#pragma unroll
        for (int j = 0; j < 4; ++j) y[i][c][j] = (lns[c][j] * rs) * (acc2[i][c][j] - mean) + lno[c][j];
        if constexpr (RESID) y[i][c] = xa[i][c] + y[i][c];
        if (valid[i]) reinterpret_cast<f32x4*>(a.nlat)[rcv[i] * 32 + 8 * w + 4 * c + g] = y[i][c];
      }
      if constexpr (PROJ) {
        uint32_t orv;
        ms_stage<false>(sB3[i], w, lane, y[i][0], y[i][1], orv, guard.big);
        guard.post(sMx[it & 1][(2) * T + (i)], orv);
      }
    }
    if (it < 2) MS_STAMP(10 + 12 * it);
    if constexpr (PROJ) {
      __syncthreads();
      if (it < 2) MS_STAMP(11 + 12 * it);
#pragma unroll
      for (int i = 0; i < T; ++i) {
        guard.tile_codes(sMx[it & 1][(2) * T + (i)], sMx[(it & 1) ^ 1][(2) * T + (i)]);
        f32x4 accp[4] = {bpv[0], bpv[1], bpv[2], bpv[3]};
        ms_gemm<4, 4>(sB3[i], lane, wph, wpl, accp);
        if (valid[i]) {
#pragma unroll
          for (int c = 0; c < 4; ++c) reinterpret_cast<f32x4*>(a.psr)[rcv[i] * 64 + 16 * w + 4 * c + g] = accp[c];
        }
      }
    }
    if constexpr (DEC) {
      // ---- decoder: hidden = relu(Wd0^T y + bd0) (this wave's two blocks), out = Wd1^T hidden (one 16-wide block,
      // computed by every wave, used by wave 0), then the integrator for the tile's 16 nodes (lanes g == 0 of wave 0)
      lb_integ_in pre[T == 1 ? 1 : 1];  // (prefetched only in the one-tile variant: the two-tile one has no registers to spare)
#pragma unroll
      for (int i = 0; i < T; ++i) {
        if constexpr (T == 1) {
          if (a.integ.on && w == 0 && g == 0 && valid[i])
            pre[0] = lb_integrate_fetch(geom, a.n_rows, a.integ.win, step, a.integ.ptype, rcv[i]);
        }
        uint32_t orv;
        ms_stage<false>(sB3[i], w, lane, y[i][0], y[i][1], orv, guard.big);
        guard.post(sMx[it & 1][(2) * T + (i)], orv);
      }
      __syncthreads();
      f32x4 hd[T][2];
#pragma unroll
      for (int i = 0; i < T; ++i) {
        guard.tile_codes(sMx[it & 1][(2) * T + (i)], sMx[(it & 1) ^ 1][(2) * T + (i)]);
        hd[i][0] = bd0v[0];
        hd[i][1] = bd0v[1];
        ms_gemm<4, 2>(sB3[i], lane, wd0h, wd0l, hd[i]);
      }
#pragma unroll
      for (int i = 0; i < T; ++i) {  // (sB2 / sMx[1] were last read before the LayerNorm barrier)
        uint32_t orv;
        ms_stage<true>(sB2[i], w, lane, hd[i][0], hd[i][1], orv, guard.big);
        guard.post(sMx[it & 1][(3) * T + (i)], orv);
      }
      __syncthreads();
#pragma unroll
      for (int i = 0; i < T; ++i) {
        guard.tile_codes(sMx[it & 1][(3) * T + (i)], sMx[(it & 1) ^ 1][(3) * T + (i)]);
        f32x4 o[1] = {f32x4{0.f, 0.f, 0.f, 0.f}};
        ms_gemm<4, 1>(sB2[i], lane, wd1h, wd1l, o);
        if (w == 0 && g == 0 && valid[i]) {
          const f32x4 b1v = *reinterpret_cast<const f32x4*>(a.bd1);
          const f32x4 acc_o = o[0] * a.dec_unscale + b1v;
          reinterpret_cast<f32x4*>(a.acc_out)[rcv[i]] = acc_o;
          bool bad = false;
          for (int d = 0; d < a.out_dim; ++d) bad |= !(fabsf(acc_o[d]) <= 3.0e38f);
          if (bad) lb_raise_math(a.ctrl, LB_MATH_NONFINITE);
          if (a.integ.on) {
            const float av[4] = {acc_o[0], acc_o[1], acc_o[2], acc_o[3]};
            lb_integrate_body(geom, a.n_rows, a.integ.win, step, a.integ.ptype, av, nullptr, a.integ.traj, a.integ.T,
                              a.integ.pred, a.integ.pred_T, rcv[i], T == 1 ? &pre[0] : nullptr);
          }
        }
      }
    }
    if (it < 2) MS_STAMP(12 + 12 * it);
  }
  MS_STAMP(30);
  guard.commit(a.ctrl, lane);
  if constexpr (DEC) {
    // the step counter is advanced by the LAST workgroup to finish (k_integrate's job)
    if (a.integ.on) {
      __syncthreads();
      if (tid == 0 && atomicAdd(a.integ.blocks_done, 1) == (int)gridDim.x - 1) {
        *a.integ.blocks_done = 0;
        const_cast<lb_ctrl*>(a.ctrl)->step = step + 1;
      }
    }
  }
  MS_STAMP(31);
}

template <int NKA, bool AGG, bool RESID, bool PROJ, int T, bool DEC = false>
__global__ void __launch_bounds__(MS_THREADS, 1) k_node_ms(lb_nms_args a, lb_geom geom) {
  using SM = ms_node_smem<NKA, AGG, T, (PROJ || DEC)>;
  __shared__ f32x4 smem[SM::f32x4_count];
  SM sm;
  sm.carve(smem);
  ms_node_body<NKA, AGG, RESID, PROJ, T, DEC>(a, geom, sm);
}

// =========================================================================================== launchers
// one tile per workgroup and iteration; two edge workgroups / one node workgroup per CU
// -DLB_MS_STAMPS builds only: every launch is followed by a device sync and a dump of workgroup 0's stamps
static long long* ms_dbg_buf() {
#ifdef LB_MS_STAMPS
  static const bool on = true;
#else
  static const bool on = false;
#endif
  static long long* buf = nullptr;
  if (on && !buf) {
    if (hipMalloc((void**)&buf, 32 * sizeof(long long)) != hipSuccess) buf = nullptr;
  }
  if (buf) (void)hipMemset(buf, 0, 32 * sizeof(long long));
  return buf;
}
static void ms_dbg_dump(const char* what, long long* buf) {
  if (!buf) return;
  long long h[32];
  (void)hipDeviceSynchronize();
  if (hipMemcpy(h, buf, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return;
  fprintf(stderr, "[ms_dbg] %s:", what);
  for (int k = 0; k < 32; ++k)
    if (h[k]) fprintf(stderr, " %d:%lld", k, h[k] - h[0]);
  fprintf(stderr, "\n");
}

static int ms_grid(int64_t tiles, int per_cu) {
  int64_t gq = (tiles + 7) / 8 * 8;  // the XCD-aware walk wants a multiple of 8
  const int64_t cap = 256 * per_cu;
  return (int)(gq < 8 ? 8 : (gq > cap ? cap : gq));
}

int lbk_edge_ms(lb_engine* e, const lb_ems_args& a_in) {
  lb_ems_args a = a_in;
  a.dbg = ms_dbg_buf();
  const int64_t tiles_cap = ((int64_t)e->e_cap * e->g.B + 15) / 16;
  const dim3 grid(ms_grid(tiles_cap, 2)), block(MS_THREADS);
  if (a.skip_elat_store)
    LB_LAUNCH_TIMED(e, (k_edge_ms<true>), grid, block, a);
  else
    LB_LAUNCH_TIMED(e, (k_edge_ms<false>), grid, block, a);
  ms_dbg_dump("edge", a.dbg);
  LB_HIP(hipGetLastError());
  return LB_OK;
}

int lbk_edge_enc_ms(lb_engine* e, const lb_ems_args& a_in) {
  lb_ems_args a = a_in;
  const int64_t tiles_cap = ((int64_t)e->e_cap * e->g.B + 15) / 16;
  hipLaunchKernelGGL(k_edge_enc_ms, dim3(ms_grid(tiles_cap, 2)), dim3(MS_THREADS), 0, e->stream, a);
  LB_HIP(hipGetLastError());
  return LB_OK;
}

int lbk_node_ms(lb_engine* e, const lb_nms_args& a_in, int nka, bool agg, bool resid, bool proj, bool dec) {
  lb_nms_args a = a_in;
  a.dbg = ms_dbg_buf();
  const int64_t tiles = (a.n_rows + 15) / 16;
  // more than one tile per CU: two tiles per iteration (both tiles' loads in flight together)
  const bool t2 = tiles > 256;
  const dim3 grid(ms_grid(t2 ? (tiles + 1) / 2 : tiles, 1)), block(MS_THREADS);
  if (dec) {  // last processor layer + decoder (+ integrator)
    if (!(nka == 4 && agg && resid && !proj)) return lb_fail(LB_ERR_UNSUPPORTED, "k_node_ms<DEC>: processor shape only");
    if (t2)
      LB_LAUNCH_TIMED(e, (k_node_ms<4, true, true, false, 2, true>), grid, block, a, e->g);
    else
      LB_LAUNCH_TIMED(e, (k_node_ms<4, true, true, false, 1, true>), grid, block, a, e->g);
    ms_dbg_dump("node+dec", a.dbg);
    LB_HIP(hipGetLastError());
    return LB_OK;
  }
#define LB_NMS(A, G, R)                                                          \
  do {                                                                           \
    if (proj && t2)                                                              \
      LB_LAUNCH_TIMED(e, (k_node_ms<A, G, R, true, 2>), grid, block, a, e->g);   \
    else if (proj)                                                               \
      LB_LAUNCH_TIMED(e, (k_node_ms<A, G, R, true, 1>), grid, block, a, e->g);   \
    else if (t2)                                                                 \
      LB_LAUNCH_TIMED(e, (k_node_ms<A, G, R, false, 2>), grid, block, a, e->g);  \
    else                                                                         \
      LB_LAUNCH_TIMED(e, (k_node_ms<A, G, R, false, 1>), grid, block, a, e->g);  \
  } while (0)
  if (nka == 4 && agg && resid)
    LB_NMS(4, true, true);
  else if (nka == 1 && !agg && !resid)
    LB_NMS(1, false, false);
  else if (nka == 2 && !agg && !resid)
    LB_NMS(2, false, false);
  else if (nka == 3 && !agg && !resid)
    LB_NMS(3, false, false);
  else if (nka == 4 && !agg && !resid)
    LB_NMS(4, false, false);
  else
    return lb_fail(LB_ERR_UNSUPPORTED, "k_node_ms<%d,%d,%d> not instantiated", nka, (int)agg, (int)resid);
#undef LB_NMS
  ms_dbg_dump("node", a.dbg);
  LB_HIP(hipGetLastError());
  return LB_OK;
}

// =========================================================================================== owner-layout kernels (round 5)
// ONE LAUNCH PER MESSAGE-PASSING LAYER for single trajectories.  The edge list is CSR by receiver,
// so a workgroup that owns the 16 receivers of node tile q also owns every edge that aggregates into them: edge MLP ->
// segment sums -> node MLP -> next projection need no kernel boundary in between, only the gathered SENDER projections
// cross workgroups, at the launch boundary (double-buffered: a layer reads psr[k & 1] and writes psr[(k + 1) & 1]).
// 24 launches per rollout step become 13.
// Owner tiles.  The 16-edge tiles of the edge latents are aligned to the node tile's own edge range instead of to the
// global edge index: node tile q (edges R0 = row_ptr[16 q] .. R1 = row_ptr[16 q + 16]) has c = ceil((R1 - R0) / 16)
// tiles, tile j holds the edges R0 + 16 j .. + 15 (the last one partly filled) and lives in slot (R0 >> 4) + q + j of
// the tile-blocked latents and of the partial-sum slots - a closed form, no scan: slots of consecutive node tiles do not
// overlap because ceil(cnt / 16) <= floor((R0 % 16 + cnt) / 16) + 1, and there are at most E / 16 + BN / 16 + 1 of them
// (lb_api.hip sizes the buffers for that).  A receiver's edges never cross a node tile's range, so the segment rule of
// the batch kernels (complete rows -> agg[r], the <= 2 segments a tile boundary cuts -> part[slot][0 | 1], combined by
// the node phase in tile order) carries over with the origin moved to R0.
// The edge phase is k_edge16v's per-tile body (the whole tile in one wave, both matrices in LDS, chained registers);
// the node phase is ms_node_body (M-split, weights in registers) in the LDS the edge weights occupied.
struct own_tiles {  // one node tile's owner tiles
  int R0, R1, c, S;
  __device__ __forceinline__ void init(const int32_t* __restrict__ row_ptr, int64_t q, int64_t n_rows) {
    const int64_t r1 = (q + 1) * 16 < n_rows ? (q + 1) * 16 : n_rows;
    R0 = row_ptr[q * 16];
    R1 = row_ptr[r1];
    c = (R1 - R0 + 15) >> 4;
    S = (R0 >> 4) + (int)q;
  }
};

// Encoder edge MLP (gns.py:73-84) into the owner layout: k_edge_enc16v's tile body, node tiles over the workgroups,
// a node tile's owner tiles over the 8 waves.
__global__ void __launch_bounds__(512, 2) k_edge_enc_own(lb_edge16_args a, const int32_t* __restrict__ row_ptr, int64_t n_rows) {
  constexpr int THREADS = 512, WAVES = 8, NW0 = 1024;
  __shared__ f32x4 sW[NW0 + 4096 + 128];  // W0 | W1 | b1 | ln scale | ln offset | b0
  const int poisoned = a.ctrl->overflow_step;
  const int tid = threadIdx.x;
  {
    const f32x4* g0 = reinterpret_cast<const f32x4*>(a.w0p);
    const f32x4* g1 = reinterpret_cast<const f32x4*>(a.w1p);
    for (int i = tid; i < NW0; i += THREADS) sW[i] = g0[i];
    for (int i = tid; i < 4096; i += THREADS) sW[NW0 + i] = g1[i];
    if (tid < 128) {
      const float* src = tid < 32 ? a.b1 : (tid < 64 ? a.ln_s : (tid < 96 ? a.ln_o : a.b0));
      sW[NW0 + 4096 + tid] = reinterpret_cast<const f32x4*>(src)[tid & 31];
    }
  }
  if (poisoned >= 0) return;
  __syncthreads();
  const float ln_inv_d = a.ctrl->ln_inv_d, ln_pad = a.ctrl->ln_pad;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, g = lane >> 4;
  ms_walk wk;
  if (!wk.init((int)((n_rows + 15) >> 4))) return;
  const lds_cptr w0b = (lds_cptr)(sW + lane), w1b = (lds_cptr)(sW + NW0 + lane);
  const lds_cptr vecb = (lds_cptr)(sW + NW0 + 4096 + g);
  const f32x4* ef4 = reinterpret_cast<const f32x4*>(a.efeat);
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  bool first = true;
  int q = wk.q;
  for (int it = 0; it < wk.n_iter; ++it, q += wk.stride) {
    own_tiles ot;
    ot.init(row_ptr, q, n_rows);
    for (int j = wave; j < ot.c; j += WAVES) {
      const int row = ot.R0 + 16 * j + n;
      const int64_t rc = row < ot.R1 ? row : ot.R1 - 1;
      const f32x4 f = ef4[rc * 2 + (g & 1)];
      const f32x4 v2[2] = {g < 2 ? f : zero, zero};
      f32x4 acc[8], acc2[8];
#pragma unroll
      for (int mb = 0; mb < 8; ++mb) acc[mb] = vecb[96 + 4 * mb];
      lb_gemm16v<false, 1>(w0b, v2, acc);
      if (first) lb_range_probe(a.ctrl, acc, 8);
      first = false;
#pragma unroll
      for (int mb = 0; mb < 8; ++mb) acc2[mb] = vecb[4 * mb];
      lb_gemm16v<true>(w1b, acc, acc2);
      f32x4 y[8];
      lb_layernorm16<true>(acc2, vecb + 32, vecb + 64, y, ln_inv_d, ln_pad);
      f32x4* ew = reinterpret_cast<f32x4*>(a.elat) + (int64_t)(ot.S + j) * 512 + lane;
#pragma unroll
      for (int mb = 0; mb < 8; ++mb) ew[64 * mb] = y[mb];
    }
  }
}

// One processor layer (gns.py:83-124) in one launch: edge phase over the owner tiles of this workgroup's node tiles
// (4 waves, a tile each), then the node phase on those node tiles.  SKIP: last layer (no edge-latent store);
// PROJ: the next layer's projection; DEC: decoder (+ integrator of a rollout step) in the epilogue.
template <bool SKIP, bool PROJ, bool DEC, int GUARD>
__global__ void __launch_bounds__(MS_THREADS, 1) k_layer_own(lb_edge16_args ea, lb_nms_args na, lb_geom geom) {
  constexpr int NW0 = 4096;
  using SM = ms_node_smem<4, true, 1, (PROJ || DEC)>;
  static_assert(SM::f32x4_count <= NW0 + 4096 + 96, "the node phase lives in the edge weights' LDS");
  __shared__ f32x4 sW[NW0 + 4096 + 96];
  const int poisoned = ea.ctrl->overflow_step;
  const float ln_inv_d = ea.ctrl->ln_inv_d, ln_pad = ea.ctrl->ln_pad;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, g = lane >> 4;
  ms_walk wk;
  const bool has_work = wk.init((int)((na.n_rows + 15) >> 4));
  SM sm;
  sm.carve(sW);
  if (poisoned >= 0 || !has_work) {  // (the node body does the idle workgroup's share of the step bookkeeping)
    ms_node_body<4, true, true, PROJ, 1, DEC, true>(na, geom, sm);
    return;
  }
  {
    const f32x4* g0 = reinterpret_cast<const f32x4*>(ea.w0p);
    const f32x4* g1 = reinterpret_cast<const f32x4*>(ea.w1p);
    for (int i = tid; i < NW0; i += MS_THREADS) sW[i] = g0[i];
    for (int i = tid; i < 4096; i += MS_THREADS) sW[NW0 + i] = g1[i];
    if (tid < 96) {
      const float* src = tid < 32 ? ea.b1 : (tid < 64 ? ea.ln_s : ea.ln_o);
      sW[NW0 + 4096 + tid] = reinterpret_cast<const f32x4*>(src)[tid & 31];
    }
  }
  __syncthreads();
  {
    uint32_t off0 = (uint32_t)(uintptr_t)(lds_cptr)(sW + lane);
    uint32_t off1 = (uint32_t)(uintptr_t)(lds_cptr)(sW + NW0 + lane);
    uint32_t off2 = (uint32_t)(uintptr_t)(lds_cptr)(sW + NW0 + 4096 + g);
    asm volatile("" : "+v"(off0), "+v"(off1), "+v"(off2));
    const lds_cptr w0b = (lds_cptr)(uintptr_t)off0, w1b = (lds_cptr)(uintptr_t)off1, vecb = (lds_cptr)(uintptr_t)off2;
    const __amdgpu_buffer_rsrc_t out_rs = __builtin_amdgcn_make_buffer_rsrc(ea.agg, 0, (int)ea.aggpart_bytes, 0x00020000);
    const uint32_t part_off = (uint32_t)((const char*)ea.part - (const char*)ea.agg);
    const f32x4* psr4 = reinterpret_cast<const f32x4*>(ea.psr);
    typedef uint32_t u32x4o __attribute__((ext_vector_type(4)));
    int guard_tiny = 0;
    bool first = true;
    int q = wk.q;
    for (int it = 0; it < wk.n_iter; ++it, q += wk.stride) {
      own_tiles ot;
      ot.init(ea.row_ptr, q, na.n_rows);
      for (int j = wave; j < ot.c; j += MS_WAVES) {
        const int e0 = ot.R0 + 16 * j, row = e0 + n;
        const bool valid = row < ot.R1;
        const int rowc = valid ? row : ot.R1 - 1;
        const int s_c = ea.senders[rowc], r_c = ea.receivers[rowc];
        // receivers of the edge just before / after the tile INSIDE the node tile's range (lane parity 0 / 1)
        int pi = (lane & 1) ? e0 + 16 : e0 - 1;
        pi = pi < ot.R0 ? ot.R0 : (pi < ot.R1 ? pi : ot.R1 - 1);
        const int rb = ea.receivers[pi];
        const int slot = ot.S + j;
        f32x4 ve[8], acc[8], p0[8];
        {
          const f32x4* er = reinterpret_cast<const f32x4*>(ea.elat) + (int64_t)slot * 512 + lane;
          const f32x4* ps = psr4 + (int64_t)s_c * 64 + g;
          const f32x4* pr = psr4 + (int64_t)r_c * 64 + 32 + g;
#pragma unroll
          for (int mb = 0; mb < 8; ++mb) {
            ve[mb] = er[64 * mb];
            acc[mb] = pr[4 * mb];
            p0[mb] = ps[4 * mb];
          }
        }
        if (first) lb_range_probe(ea.ctrl, ve, 8);
        uint32_t or_e = 0, or_h = 0;
        lb_gemm16v<false, 4, GUARD>(w0b, ve, acc, &or_e);
#pragma unroll
        for (int mb = 0; mb < 8; ++mb) acc[mb] = lb_pk_add(acc[mb], p0[mb]);
        if (first) lb_range_probe(ea.ctrl, acc, 8);
        first = false;
        f32x4 acc2[8];
#pragma unroll
        for (int mb = 0; mb < 8; ++mb) acc2[mb] = vecb[4 * mb];
        lb_gemm16v<true, 4, GUARD>(w1b, acc, acc2, &or_h);
        if constexpr (GUARD != 0) guard_tiny |= (int)lb_rows_tiny(or_e) | (int)lb_rows_tiny(or_h);
        f32x4 y[8];
        lb_layernorm16<true>(acc2, vecb + 32, vecb + 64, y, ln_inv_d, ln_pad);
        if constexpr (!SKIP) {
          f32x4* ew = reinterpret_cast<f32x4*>(ea.elat) + (int64_t)slot * 512 + lane;
#pragma unroll
          for (int mb = 0; mb < 8; ++mb) ew[64 * mb] = lb_pk_add(ve[mb], y[mb]);
        }
        // fused jraph.segment_sum: segmented scan inside each 16-lane DPP row (rows past R1: copies of the node tile's
        // last edge, each its own segment, above every valid lane, never stored)
        const int rr = valid ? r_c : (-1 - n);
        const int r_prev = __builtin_amdgcn_update_dpp(-2, rr, 0x111, 0xF, 0xF, false);
        const bool head = (n == 0) || (rr != r_prev);
        const unsigned H = (unsigned)(__ballot(head) & 0xffffull);
        const int segstart = 31 - __clz(H & ((2u << n) - 1u));
        const bool tail = (n == 15) || ((H >> (n + 1)) & 1u);
        const float m1 = (n >= 1 && segstart <= n - 1) ? 1.f : 0.f, m2 = (n >= 2 && segstart <= n - 2) ? 1.f : 0.f;
        const float m4 = (n >= 4 && segstart <= n - 4) ? 1.f : 0.f, m8 = (n >= 8 && segstart <= n - 8) ? 1.f : 0.f;
#pragma unroll
        for (int mb = 0; mb < 8; mb += 2) lb_scan8(y[mb], y[mb + 1], m1, m2, m4, m8);
        {
          const int r_before = __builtin_amdgcn_readlane(rb, 0), r_after = __builtin_amdgcn_readlane(rb, 1);
          const bool starts_before = segstart == 0 && j > 0 && r_before == rr;
          const bool ends_after = n == 15 && e0 + 16 < ot.R1 && r_after == rr;
          const int slot01 = segstart == 0 ? 0 : 1;
          const uint32_t row_off = (!starts_before && !ends_after) ? (uint32_t)rr * 512u
                                                                  : part_off + ((uint32_t)slot * 2u + (uint32_t)slot01) * 512u;
          const uint32_t off = (tail && valid) ? row_off + (uint32_t)g * 16u : 0x80000000u;  // out of range: dropped
#pragma unroll
          for (int mb = 0; mb < 8; ++mb)
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4o, y[mb]), out_rs, (int)off, 64 * mb, 0);
        }
      }
    }
    if (guard_tiny && lane == 0) lb_raise_math(ea.ctrl, LB_MATH_TINY);
  }
  // the aggregates of this workgroup's node tiles are complete: every wave's stores are visible to the workgroup behind
  // the barrier (workgroup-scope release / acquire), and the edge weights' LDS is free for the node phase
  __syncthreads();
  ms_node_body<4, true, true, PROJ, 1, DEC, true>(na, geom, sm);
}

int lbk_edge_enc_own(lb_engine* e, const lb_edge16_args& a) {
  const int64_t tiles = (e->BN + 15) / 16;
  hipLaunchKernelGGL(k_edge_enc_own, dim3(ms_grid(tiles, 2)), dim3(512), 0, e->stream, a, e->row_ptr, e->BN);
  LB_HIP(hipGetLastError());
  return LB_OK;
}

int lbk_layer_own(lb_engine* e, const lb_edge16_args& ea, const lb_nms_args& na_in, bool proj, bool dec) {
  lb_nms_args na = na_in;
  na.dbg = nullptr;
  const int64_t tiles = (na.n_rows + 15) / 16;
  const dim3 grid(ms_grid(tiles, 1)), block(MS_THREADS);
  const bool guard = e->math_auto && !e->guard_sampled;
  if (dec && proj) return lb_fail(LB_ERR_ARG, "k_layer_own: the decoder follows the last layer");
#define LB_OWN_GO(SK, PR, DE)                                                            \
  do {                                                                                   \
    if (guard)                                                                           \
      LB_LAUNCH_TIMED(e, (k_layer_own<SK, PR, DE, 1>), grid, block, ea, na, e->g);       \
    else                                                                                 \
      LB_LAUNCH_TIMED(e, (k_layer_own<SK, PR, DE, 0>), grid, block, ea, na, e->g);       \
  } while (0)
  if (proj) {
    if (ea.skip_elat_store) LB_OWN_GO(true, true, false); else LB_OWN_GO(false, true, false);
  } else if (dec) {
    if (ea.skip_elat_store) LB_OWN_GO(true, false, true); else LB_OWN_GO(false, false, true);
  } else {
    if (ea.skip_elat_store) LB_OWN_GO(true, false, false); else LB_OWN_GO(false, false, false);
  }
#undef LB_OWN_GO
  LB_HIP(hipGetLastError());
  return LB_OK;
}
