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
template <int NKA, bool AGG, bool RESID, bool PROJ, int T, bool DEC = false>
__global__ void __launch_bounds__(MS_THREADS, 1) k_node_ms(lb_nms_args a, lb_geom geom) {
  static_assert(!(DEC && PROJ), "the decoder follows the LAST layer");
  constexpr int NK0 = NKA + (AGG ? 4 : 0);
  __shared__ f32x4 sB1[T][NK0 * 2 * 64];
  __shared__ f32x4 sB2[T][4 * 2 * 64];
  __shared__ f32x4 sB3[(PROJ || DEC) ? T : 1][4 * 2 * 64];
  __shared__ __attribute__((aligned(16))) f32x2m sRed[T][16 * 4];
  __shared__ __attribute__((aligned(16))) int sMx[2][4 * T][MS_GUARD_WORDS];
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
          const int t0 = k0[i] >> 4, t1 = (k1[i] - 1) >> 4;
          const bool single = t0 == t1;
          const int nsrc = (k1[i] <= k0[i]) ? 0 : (single ? 1 : t1 - t0 + 1);
          const int kk0 = k0[i];
          auto slot_of = [&](int tt) -> const f32x4* {
            const float* src = single ? a.agg + rc * 128 : a.part + ((int64_t)tt * 2 + (kk0 <= (tt << 4) ? 0 : 1)) * 128;
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
  for (int i = tid; i < (int)(sizeof(sMx) / sizeof(int)); i += MS_THREADS) reinterpret_cast<int*>(sMx)[i] = 0;
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
