// lb_edge16v.hip - round-2 rewrite of the processor edge kernel (f16x2, fused segment_sum).
//
// Same mathematics, layouts and tile walk as k_edge16n in lb_edge16.hip (reference:
// GNS._processor update_edge_features + jraph.segment_sum, models/gns.py:86-122); what changes is
// the instruction stream.  Round-1 profile of k_edge16n (TGV3D-8k x 8): compute-only 205 us, memory
// only 263 us, together 308 us - the matrix pipe itself is busy 86 us, the rest of the compute time
// is VALU issue slots and LDS round trips that sit in series with the MFMAs:
//   * 84 v_or_b32 per tile only to form LDS addresses past the 64 KiB ds offset field,
//   * v_mov_b32_dpp + v_pk_fma_f32 pairs in the segmented scan (SLP-packed fmas cannot take a DPP
//     source; packed fp32 VALU is slower beside MFMAs than two plain ones),
//   * 20 VALU per fp16 hi/lo split of 8 values, a canonicalising v_max pair per ReLU,
//   * under the 168-VGPR cap of three waves per SIMD the compiler serialised
//     ds_read -> s_waitcnt lgkmcnt(0) -> v_mfma in the second GEMM (one LDS round trip per MFMA).
// Here:
//   * two lane bases (W0 image, W1 image) keep every ds_read offset inside the 16-bit field;
//   * the split is v_cvt_pk_f16_f32 + v_fma_mixlo/mixhi_f16 (12 VALU per 8 values), ReLU is an
//     integer max, the scan is v_fmac_f32_dpp in fixed-order asm blocks (128 VALU per tile);
//   * the GEMM is phase-pipelined with two 16-register fragment buffers: the `lo` fragments of
//     block k+1 are fetched before the eight `hi` MFMAs of block k issue and the `hi` fragments
//     before its four `lo` MFMAs, every accumulator is touched again only after four independent
//     MFMAs (accumulate-chain spacing), phases are pinned with sched_barrier;
//   * RELOAD: the edge latents are not kept in registers for the residual but read a second time
//     (L2 / Infinity-Cache hit ~2 us after the first read) - the kernel then fits 128 VGPRs, i.e.
//     FOUR waves per SIMD (one 1024-thread workgroup per CU) to hide HBM latency.
#include <stdlib.h>

#include "lb_f16x2.h"

// ABL (tools/edge_ab.hip only, 0 in the product): 1 no psr gathers, 2 no edge-latent loads,
// 4 no stores, 8 no GEMMs, 16 no LayerNorm / scan (epilogue VALU).
// SKIP: last processor layer - the updated edge latents have no reader (compile-time so that the
// residual path is branch-free: a store under a branch costs a vmcnt(0) drain at the join).
// NT: the edge latents are streamed with nontemporal loads / stores (batches whose latents exceed the 256 MiB
// Infinity Cache); a single trajectory's latents (tens of MB) are read back from the cache by the next layer, so
// small graphs use plain accesses.
// GUARD: 1 (default in guarded mode, round 4) per-ROW TINY test on one k-group of both GEMM operands of every tile
// (lb_rows_tiny); 2 the same on every k-group plus the tile-wide test (+8 % kernel time: lb_math_mode 3 / LB_GUARD=full);
// 0 only the sampled probe (first tile of every wave: LB_GUARD=sampled, or unguarded arithmetic).
// TICKET (round 4): the waves of a workgroup draw their tiles from an LDS ticket counter inside the workgroup's contiguous
// chunk instead of walking a static stride - the SIMD's issue arbiter prefers its oldest wave, which otherwise finishes its
// share early and leaves the younger wave to run alone at the end of the launch (k_sg_msg: lb_segnn_msg.hip).
template <int WPS, bool RELOAD, bool SKIP, int ABL = 0, bool PRIO = false, bool NT = true, int GUARD = 0, bool TICKET = false>
__global__ void __launch_bounds__(WPS * 256, WPS) k_edge16v(lb_edge16_args a) {
  constexpr int THREADS = WPS * 256, WAVES = WPS * 4;
  constexpr int NW0 = 4096;
  __shared__ f32x4 sW[NW0 + 4096 + 96];
  __shared__ int s_ticket;
  // Prologue order (matters for small graphs, where a launch is a latency chain): the control block is read,
  // the weight loads are issued into registers, the first tile's indices are requested while those are in
  // flight, and only then the weights are written to LDS - "flag -> weights -> barrier -> indices -> gathers" was
  // four dependent round trips, this is three.  The poison flag is acted on before anything is stored; the
  // loads issued before that are in bounds whatever the state (n_edges_total is clamped to the allocation).
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
  // the wave index is uniform: keep the whole tile walk (t, stride, bounds) in scalar registers
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, g = lane >> 4;
  const int xcd = blockIdx.x & 7;
  int stride = (gridDim.x >> 3) * WAVES;
  int t_lo = (int)(((int64_t)ntiles * xcd) >> 3), t_hi = (int)(((int64_t)ntiles * (xcd + 1)) >> 3);
  int t = t_lo + (blockIdx.x >> 3) * WAVES + wave;
  if constexpr (TICKET) {  // ticket k of workgroup b of the XCD -> tile t_lo + b + k * (workgroups per XCD): the workgroups
    // still stream through the XCD's range side by side (contiguous chunks per workgroup measured 3 % slower than the
    // static walk: 32 separate streams per XCD); first tile static
    stride = gridDim.x >> 3;
    t_lo += (blockIdx.x >> 3);
    t = t_lo + wave * stride;
  }
  auto rowc_of = [&](int tt) -> int64_t {
    const int row = tt * 16 + n;
    return row < E ? row : (E > 0 ? E - 1 : 0);
  };
  int s_c = 0, r_c = 0;
  if (t < t_hi) {
    const int64_t rc = rowc_of(t);
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
  // two lane bases so that every fragment offset fits the 16-bit ds offset field; the integer
  // round trip through an asm keeps the compiler from folding them back into one base + 64 KiB
  uint32_t off0 = (uint32_t)(uintptr_t)(lds_cptr)(sW + lane);
  uint32_t off1 = (uint32_t)(uintptr_t)(lds_cptr)(sW + NW0 + lane);
  uint32_t off2 = (uint32_t)(uintptr_t)(lds_cptr)(sW + NW0 + 4096 + g);
  asm volatile("" : "+v"(off0), "+v"(off1), "+v"(off2));
  const lds_cptr w0b = (lds_cptr)(uintptr_t)off0, w1b = (lds_cptr)(uintptr_t)off1, vecb = (lds_cptr)(uintptr_t)off2;
  const f32x4* psr4 = reinterpret_cast<const f32x4*>(a.psr);
  const int n_iter = TICKET ? 0x7fffffff : (t_hi - 1 - t) / stride + 1;
  const int t_last = TICKET ? t_hi - 1 : t + (n_iter - 1) * stride;
  // (the first tile's indices were waited for above, by the barrier: a wait at the loop header would also be
  // executed on the back edge, where it drains the previous tile's stores)
  asm volatile("" : "+v"(s_c), "+v"(r_c));
  int guard_tiny = 0;  // exhaustive TINY guard: every tile's two GEMM operands (lb_tile_tiny)
  for (int it = 0; it < n_iter && t < t_hi; ++it) {
    int t_next = t + stride;
    if constexpr (TICKET) {  // the next tile's ticket: one lane draws, the wave reads it back as a scalar
      int k = 0;
      if (lane == 0) k = atomicAdd(&s_ticket, 1);
      t_next = t_lo + __builtin_amdgcn_readfirstlane(k) * stride;
    }
    f32x4 acc[8], ve[8];
    const int r_cur = r_c;
    const f32x4* er = reinterpret_cast<const f32x4*>(a.elat) + (int64_t)t * 512 + lane;
    {
      const f32x4* ps = psr4 + (int64_t)s_c * 64 + g;
      const f32x4* pr = psr4 + (int64_t)r_c * 64 + 32 + g;
      f32x4 p0[8];
#pragma unroll
      for (int mb = 0; mb < 8; ++mb) {
        // streamed once per layer: nontemporal unless the RELOAD variant wants the tile back from L2
        ve[mb] = (ABL & 2) ? f32x4{1.f, 2.f, (float)(t & 1023), (float)mb}
                           : ((RELOAD || !NT) ? er[64 * mb] : __builtin_nontemporal_load(&er[64 * mb]));
        p0[mb] = (ABL & 1) ? f32x4{.1f, .2f, (float)(s_c & 255), (float)mb} : ps[4 * mb];
        acc[mb] = (ABL & 1) ? f32x4{.3f, .1f, (float)(r_c & 255), (float)mb} : pr[4 * mb];
      }
      const int64_t rn = rowc_of(min(t_next, t_last));
      s_c = a.senders[rn];
      r_c = a.receivers[rn];
#pragma unroll
      for (int mb = 0; mb < 8; ++mb) acc[mb] = lb_pk_add(acc[mb], p0[mb]);
    }
    // receivers of the edge just before and just after this tile (lane parity 0 / 1): tell whether a
    // segment is cut by the tile boundary without touching row_ptr; fetched with the tile's other
    // loads for the same reason the next indices are (see below)
    int rb = lb_edge_probe(a.receivers, t, lane, E);
    if (it == 0) lb_range_probe(a.ctrl, ve, 8);  // f16x2 range guard, first tile of every wave
    // PRIO: a wave in its GEMM phase outranks the waves of the SIMD that are in their VALU epilogue: its
    // MFMAs then issue back to back and the others fill the issue slots in between (VALU beside a busy
    // matrix pipe still runs at ~1 instruction per 7 cycles, tools/simd_overlap_bench)
    if constexpr (PRIO) __builtin_amdgcn_s_setprio(2);
    uint32_t or_e = 0, or_h = 0;
    if constexpr (!(ABL & 8)) lb_gemm16v<false, 4, GUARD>(w0b, ve, acc, &or_e);
    if (it == 0) lb_range_probe(a.ctrl, acc, 8);
    f32x4 acc2[8];
#pragma unroll
    for (int mb = 0; mb < 8; ++mb) acc2[mb] = vecb[4 * mb];
    if constexpr (!(ABL & 8)) {
      lb_gemm16v<true, 4, GUARD>(w1b, acc, acc2, &or_h);
      if constexpr (PRIO) __builtin_amdgcn_s_setprio(0);
      if constexpr (GUARD == 2) guard_tiny |= (int)lb_tile_tiny(or_e) | (int)lb_tile_tiny(or_h);
      if constexpr (GUARD != 0) guard_tiny |= (int)lb_rows_tiny(or_e) | (int)lb_rows_tiny(or_h);
    } else {
#pragma unroll
      for (int mb = 0; mb < 8; ++mb)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc2[mb][j] += acc[mb][j] + ve[mb][j];
    }
    if constexpr (RELOAD && !SKIP && !(ABL & 2)) {
      // second read of the edge latents for the residual (first read ~2 us ago: L2 / MALL resident)
#pragma unroll
      for (int mb = 0; mb < 8; ++mb) ve[mb] = er[64 * mb];
    }
    // take delivery of the next tile's indices HERE, while only loads are in flight: with stores
    // pending too, gfx9's single vmcnt makes any later wait a full drain (vmcnt(0)) - at the loop
    // top that would put the store latency of this tile in front of the next tile's loads
    asm volatile("" : "+v"(s_c), "+v"(r_c), "+v"(rb));
    f32x4 y[8];
    lb_layernorm16<!(ABL & 16)>(acc2, vecb + 32, vecb + 64, y, ln_inv_d, ln_pad);
    const int row = t * 16 + n;
    const bool valid = row < E;
    if constexpr (!SKIP && !(ABL & 4)) {
      f32x4* ew = reinterpret_cast<f32x4*>(a.elat_out ? a.elat_out : a.elat) + (int64_t)t * 512 + lane;
#pragma unroll
      for (int mb = 0; mb < 8; ++mb) {
        if constexpr (NT)
          __builtin_nontemporal_store(lb_pk_add(ve[mb], y[mb]), &ew[64 * mb]);
        else
          ew[64 * mb] = lb_pk_add(ve[mb], y[mb]);
      }
    }
    // fused jraph.segment_sum: segmented Hillis-Steele scan inside each 16-lane DPP row
    const int rr = valid ? r_cur : (-1 - n);
    const int r_prev = __builtin_amdgcn_update_dpp(-2, rr, 0x111, 0xF, 0xF, false);
    const bool head = (n == 0) || (rr != r_prev);
    const unsigned H = (unsigned)(__ballot(head) & 0xffffull);
    const unsigned below = H & ((2u << n) - 1u);
    const int segstart = 31 - __clz(below);
    const bool tail = (n == 15) || ((H >> (n + 1)) & 1u);
    const float m1 = (n >= 1 && segstart <= n - 1) ? 1.f : 0.f, m2 = (n >= 2 && segstart <= n - 2) ? 1.f : 0.f;
    const float m4 = (n >= 4 && segstart <= n - 4) ? 1.f : 0.f, m8 = (n >= 8 && segstart <= n - 8) ? 1.f : 0.f;
    if (!valid) {
#pragma unroll
      for (int mb = 0; mb < 8; ++mb) y[mb] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if constexpr (!(ABL & 16))
#pragma unroll
    for (int mb = 0; mb < 8; mb += 2) lb_scan8(y[mb], y[mb + 1], m1, m2, m4, m8);
    if (tail && valid && !(ABL & 4)) {
      int slot01;
      const bool complete = lb_seg_complete(rb, rr, segstart, n, t, E, slot01);
      float* dst = complete ? a.agg + (int64_t)rr * 128 : a.part + ((int64_t)t * 2 + slot01) * 128;
      f32x4* d4 = reinterpret_cast<f32x4*>(dst) + g;
#pragma unroll
      for (int mb = 0; mb < 8; ++mb) d4[4 * mb] = y[mb];
    }
    t = t_next;
  }
  if (guard_tiny && lane == 0) lb_raise_math(a.ctrl, LB_MATH_TINY);
}

// ---------------------------------------------------------------------------------------------
// k_edge_enc16v: the ENCODER edge MLP (gns.py:73-84), e0 = LayerNorm(W1 relu(W0 f + b0) + b1) over the
// edge features f = (rel_disp, rel_dist) the neighbor search wrote (8 floats per edge, zero padded).
// Write-bound: 32 B read and 512 B written per edge.  Same tile walk, block loop, LayerNorm and
// nontemporal tile-blocked store as k_edge16v; the first Linear is ONE k-step of 32 (4 KiB of W0 in LDS);
// no gathers, no residual, no aggregation - ~100 VGPRs, WPS waves per SIMD.  The next tile's features
// are fetched one tile ahead and taken delivery of before this tile's stores (in-order vmcnt).
template <int WPS>
__global__ void __launch_bounds__(WPS * 256, WPS) k_edge_enc16v(lb_edge16_args a) {
  constexpr int THREADS = WPS * 256, WAVES = WPS * 4;
  constexpr int NW0 = 1024;
  __shared__ f32x4 sW[NW0 + 4096 + 128];  // W0 | W1 | b1 | ln scale | ln offset | b0
  const int poisoned = a.ctrl->overflow_step;  // acted on after the staging loads are in flight
  const int E = a.ctrl->n_edges_total;
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
  const int ntiles = (E + 15) >> 4;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, g = lane >> 4;
  const int xcd = blockIdx.x & 7, slot = (blockIdx.x >> 3) * WAVES + wave;
  const int stride = (gridDim.x >> 3) * WAVES;
  const int t_lo = (int)(((int64_t)ntiles * xcd) >> 3), t_hi = (int)(((int64_t)ntiles * (xcd + 1)) >> 3);
  int t = t_lo + slot;
  if (t >= t_hi) return;
  const lds_cptr w0b = (lds_cptr)(sW + lane), w1b = (lds_cptr)(sW + NW0 + lane);
  const lds_cptr vecb = (lds_cptr)(sW + NW0 + 4096 + g);
  const f32x4* ef4 = reinterpret_cast<const f32x4*>(a.efeat);
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  auto feat_of = [&](int tt) -> f32x4 {
    const int row = tt * 16 + n;
    const int64_t rc = row < E ? row : E - 1;
    return ef4[rc * 2 + (g & 1)];
  };
  const int n_iter = (t_hi - 1 - t) / stride + 1;
  const int t_last = t + (n_iter - 1) * stride;
  f32x4 f_n = feat_of(t);
  asm volatile("" : "+v"(f_n));
  for (int it = 0; it < n_iter; ++it, t += stride) {
    const f32x4 v2[2] = {g < 2 ? f_n : zero, zero};
    f_n = feat_of(min(t + stride, t_last));
    f32x4 acc[8], acc2[8];
#pragma unroll
    for (int mb = 0; mb < 8; ++mb) acc[mb] = vecb[96 + 4 * mb];
    lb_gemm16v<false, 1>(w0b, v2, acc);
    if (it == 0) lb_range_probe(a.ctrl, acc, 8);
#pragma unroll
    for (int mb = 0; mb < 8; ++mb) acc2[mb] = vecb[4 * mb];
    lb_gemm16v<true>(w1b, acc, acc2);
    asm volatile("" : "+v"(f_n));
    f32x4 y[8];
    lb_layernorm16<true>(acc2, vecb + 32, vecb + 64, y, ln_inv_d, ln_pad);
    f32x4* ew = reinterpret_cast<f32x4*>(a.elat) + (int64_t)t * 512 + lane;
#pragma unroll
    for (int mb = 0; mb < 8; ++mb) __builtin_nontemporal_store(y[mb], &ew[64 * mb]);
  }
}

int lbk_edge_enc16v(lb_engine* e, const lb_edge16_args& a) {
  const int64_t tiles_cap = ((int64_t)e->e_cap * e->g.B + 15) / 16;
  auto grid_for = [&](int waves_per_block) {
    int64_t g = (tiles_cap + waves_per_block - 1) / waves_per_block;
    g = (g + 7) / 8 * 8;
    return (int)(g < 8 ? 8 : (g > 256 ? 256 : g));
  };
  if (tiles_cap <= 256 * 8 * 2)
    hipLaunchKernelGGL((k_edge_enc16v<2>), dim3(grid_for(8)), dim3(512), 0, e->stream, a);
  else
    hipLaunchKernelGGL((k_edge_enc16v<4>), dim3(256), dim3(1024), 0, e->stream, a);
  LB_HIP(hipGetLastError());
  return LB_OK;
}

// Two waves per SIMD, GEMM-phase priority (round 2's measured best; the software-prefetching, second-read and
// three- / four-wave variants were measured in round 2 - profiles/r02_edge16v_bench.txt - and are gone from the tree).
// (round 5: FOUR waves per SIMD with the second read of the latents - k_edge16v<4, true, ..>, 128 VGPRs - measured on ONE
// trajectory, where a wave has 3 - 4 tiles and fewer rounds might have paid: TGV3D-8k B = 1 42.4 us per launch against 36.4,
// 0.668 vs 0.615 ms per step; not kept.)
int lbk_edge16v(lb_engine* e, const lb_edge16_args& a) {
  // LDS tile tickets (round 4) on ONE trajectory (a few tiles per wave: the faster - older - wave of a SIMD takes more of
  // them; LDC3D-8k B = 1 0.619 -> 0.604, TGV3D-8k 0.673 -> 0.663 ms/step), the static strided walk on batches (tickets
  // neutral there: profiles/r04_ab_edge_ticket.txt).  The same size (12288 capacity tiles = 96 MiB of latents) decides
  // between plain accesses (cache-resident between layers) and nontemporal streams, so two of the four combinations exist.
  const bool small = ((int64_t)e->e_cap * e->g.B + 15) / 16 < 12288;
#define LB_E16V__(NT, G, GU, TK)                                                                        \
  do {                                                                                                  \
    if (a.skip_elat_store)                                                                              \
      LB_LAUNCH_TIMED(e, (k_edge16v<2, false, true, 0, true, NT, GU, TK>), dim3(G), dim3(512), a);      \
    else                                                                                                \
      LB_LAUNCH_TIMED(e, (k_edge16v<2, false, false, 0, true, NT, GU, TK>), dim3(G), dim3(512), a);     \
  } while (0)
#define LB_E16V(NT, G, TK)                       \
  do {                                           \
    if (e->guard_full)                           \
      LB_E16V__(NT, G, 2, TK);                   \
    else if (e->math_auto && !e->guard_sampled)  \
      LB_E16V__(NT, G, 1, TK);                   \
    else                                         \
      LB_E16V__(NT, G, 0, TK);                   \
  } while (0)
  // Small graphs (one 2.5 k-particle trajectory = ~1000 tiles): a launch is the latency chain
  // "stage 133 KiB of weights -> one tile per wave", so launch no more workgroups than there are tiles for.
  // The tile count is bounded on the host by the frozen capacity (the real count lives on the device).
  const int64_t tiles_cap = ((int64_t)e->e_cap * e->g.B + 15) / 16;
  int64_t g = (tiles_cap + 7) / 8;
  g = (g + 7) / 8 * 8;  // the XCD-aware walk wants a multiple of 8
  const int grid = (int)(g < 8 ? 8 : (g > 256 ? 256 : g));
  if (small)
    LB_E16V(false, grid, true);
  else
    LB_E16V(true, grid, false);
#undef LB_E16V
#undef LB_E16V__
  LB_HIP(hipGetLastError());
  return LB_OK;
}
