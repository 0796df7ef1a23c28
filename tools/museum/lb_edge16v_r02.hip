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

// ABL (tools/edge16v_bench.hip only, 0 in the product): 1 no psr gathers, 2 no edge-latent loads,
// 4 no stores, 8 no GEMMs, 16 no LayerNorm / scan (epilogue VALU).
// SKIP: last processor layer - the updated edge latents have no reader (compile-time so that the
// residual path is branch-free: a store under a branch costs a vmcnt(0) drain at the join).
template <int WPS, bool RELOAD, bool SKIP, int ABL = 0, bool PRIO = false>
__global__ void __launch_bounds__(WPS * 256, WPS) k_edge16v(lb_edge16_args a) {
  constexpr int THREADS = WPS * 256, WAVES = WPS * 4;
  constexpr int NW0 = 4096;
  __shared__ f32x4 sW[NW0 + 4096 + 96];
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
  const int xcd = blockIdx.x & 7, slot = (blockIdx.x >> 3) * WAVES + wave;
  const int stride = (gridDim.x >> 3) * WAVES;
  const int t_lo = (int)(((int64_t)ntiles * xcd) >> 3), t_hi = (int)(((int64_t)ntiles * (xcd + 1)) >> 3);
  int t = t_lo + slot;
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
  const int n_iter = (t_hi - 1 - t) / stride + 1;
  const int t_last = t + (n_iter - 1) * stride;
  // (the first tile's indices were waited for above, by the barrier: a wait at the loop header would also be
  // executed on the back edge, where it drains the previous tile's stores)
  asm volatile("" : "+v"(s_c), "+v"(r_c));
  for (int it = 0; it < n_iter; ++it, t += stride) {
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
        ve[mb] = (ABL & 2) ? f32x4{1.f, 2.f, (float)t, (float)mb}
                           : (RELOAD ? er[64 * mb] : __builtin_nontemporal_load(&er[64 * mb]));
        p0[mb] = (ABL & 1) ? f32x4{.1f, .2f, (float)s_c, (float)mb} : ps[4 * mb];
        acc[mb] = (ABL & 1) ? f32x4{.3f, .1f, (float)r_c, (float)mb} : pr[4 * mb];
      }
      const int64_t rn = rowc_of(min(t + stride, t_last));
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
    if constexpr (!(ABL & 8)) lb_gemm16v<false>(w0b, ve, acc);
    if (it == 0) lb_range_probe(a.ctrl, acc, 8);
    f32x4 acc2[8];
#pragma unroll
    for (int mb = 0; mb < 8; ++mb) acc2[mb] = vecb[4 * mb];
    if constexpr (!(ABL & 8)) {
      lb_gemm16v<true>(w1b, acc, acc2);
      if constexpr (PRIO) __builtin_amdgcn_s_setprio(0);
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
      for (int mb = 0; mb < 8; ++mb) __builtin_nontemporal_store(lb_pk_add(ve[mb], y[mb]), &ew[64 * mb]);
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
  }
}

// ---------------------------------------------------------------------------------------------
// k_edge16l: LATE prefetch.  Like k_edge16p every load of tile t+1 is issued by tile t, but only AFTER
// its second GEMM, when the GEMM working set (first-layer activations, both fragment buffers) is dead:
// the 96 prefetch registers never coexist with it, so two waves per SIMD keep ~90 VGPRs of slack for
// the compiler's scheduling instead of running at the 256-register cap (k_edge16p: 341 us, spill-prone).
// The loads still precede the tile's stores in program order, so the wait at the next tile's top does
// not have to cover a store acknowledgement (gfx9's single in-order vmcnt).
// (header of k_edge16p, the early-prefetch variant:)
// k_edge16p: the same tile body, fully software-pipelined inside each wave.  While tile t is in the
// GEMMs, ALL loads of tile t+1 (edge latents + gathered sender/receiver projections + CSR bounds) and
// the indices of tile t+2 are in flight, so the bytes a CU has outstanding no longer depend on how
// the phases of its waves happen to line up (fine-grained arbitration keeps equal waves in lock step:
// they all load, then all compute).  The prefetch sets cost 96 VGPRs -> WPS = 2 (256 VGPRs).
// The edge latents are touched once per layer (562 MB >> L2 + Infinity Cache): their loads and stores
// are nontemporal so that the streams do not push the gathered psr table out of L2 / MALL
// (tools/stream_bench: 257 -> 233 us for this access pattern).
// The first tile is peeled: a waitcnt at the loop header serves the entry and the back edge with ONE
// static count, and the entry's smaller count would drain the previous tile's stores on every trip.
struct lb_e16l_state {
  f32x4 ve[8], ps[8], pr[8];  // next tile, in flight
  int s_n, r_n;               // indices of the tile after next
  int r_pref, rb;             // receiver of the next tile's lanes + its boundary probe
};

template <int WPS, bool SKIP, int ABL = 0, bool PRIO = true>
__global__ void __launch_bounds__(WPS * 256, WPS) k_edge16l(lb_edge16_args a) {
  constexpr int THREADS = WPS * 256, WAVES = WPS * 4;
  constexpr int NW0 = 4096;
  __shared__ f32x4 sW[NW0 + 4096 + 96];
  if (a.ctrl->overflow_step >= 0) return;
  const int tid = threadIdx.x;
  {
    const f32x4* g0 = reinterpret_cast<const f32x4*>(a.w0p);
    const f32x4* g1 = reinterpret_cast<const f32x4*>(a.w1p);
    for (int i = tid; i < NW0; i += THREADS) sW[i] = g0[i];
    for (int i = tid; i < 4096; i += THREADS) sW[NW0 + i] = g1[i];
    if (tid < 96) {
      const float* src = tid < 32 ? a.b1 : (tid < 64 ? a.ln_s : a.ln_o);
      sW[NW0 + 4096 + tid] = reinterpret_cast<const f32x4*>(src)[tid & 31];
    }
  }
  __syncthreads();
  const int E = a.ctrl->n_edges_total;
  const float ln_inv_d = a.ctrl->ln_inv_d, ln_pad = a.ctrl->ln_pad;
  const int ntiles = (E + 15) >> 4;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, g = lane >> 4;
  const int xcd = blockIdx.x & 7, slot = (blockIdx.x >> 3) * WAVES + wave;
  const int stride = (gridDim.x >> 3) * WAVES;
  const int t_lo = (int)(((int64_t)ntiles * xcd) >> 3), t_hi = (int)(((int64_t)ntiles * (xcd + 1)) >> 3);
  int t = t_lo + slot;
  if (t >= t_hi) return;
  uint32_t off0 = (uint32_t)(uintptr_t)(lds_cptr)(sW + lane);
  uint32_t off1 = (uint32_t)(uintptr_t)(lds_cptr)(sW + NW0 + lane);
  uint32_t off2 = (uint32_t)(uintptr_t)(lds_cptr)(sW + NW0 + 4096 + g);
  asm volatile("" : "+v"(off0), "+v"(off1), "+v"(off2));
  const lds_cptr w0b = (lds_cptr)(uintptr_t)off0, w1b = (lds_cptr)(uintptr_t)off1, vecb = (lds_cptr)(uintptr_t)off2;
  auto rowc_of = [&](int tt) -> int64_t {
    const int row = tt * 16 + n;
    return row < E ? row : E - 1;
  };
  const f32x4* psr4 = reinterpret_cast<const f32x4*>(a.psr);
  const int n_iter = (t_hi - 1 - t) / stride + 1;
  const int t_last = t + (n_iter - 1) * stride;

  lb_e16l_state st;
  // every load is unconditional (tile indices clamp to the wave's last tile): a load under a
  // branch makes the compiler wait vmcnt(0) for the loop-carried registers
  auto issue = [&](int tt, int s, int r) {
    const f32x4* er = reinterpret_cast<const f32x4*>(a.elat) + (int64_t)tt * 512 + lane;
    const f32x4* ps = psr4 + (int64_t)s * 64 + g;
    const f32x4* pr = psr4 + (int64_t)r * 64 + 32 + g;
#pragma unroll
    for (int mb = 0; mb < 8; ++mb) {
      st.ve[mb] = (ABL & 2) ? f32x4{1.f, 2.f, (float)tt, (float)mb} : __builtin_nontemporal_load(&er[64 * mb]);
      st.ps[mb] = (ABL & 1) ? f32x4{.1f, .2f, (float)s, (float)mb} : ps[4 * mb];
      st.pr[mb] = (ABL & 1) ? f32x4{.3f, .1f, (float)r, (float)mb} : pr[4 * mb];
    }
    st.r_pref = r;
    st.rb = lb_edge_probe(a.receivers, tt, lane, E);
  };
  {
    const int64_t rc = rowc_of(t);
    int s0 = a.senders[rc], r0 = a.receivers[rc];
    asm volatile("" : "+v"(s0), "+v"(r0));
    issue(t, s0, r0);
    const int64_t rn = rowc_of(min(t + stride, t_last));
    st.s_n = a.senders[rn];
    st.r_n = a.receivers[rn];
  }

  auto body = [&](int tc, bool first) {
    // ---- take delivery of the prefetched tile
    f32x4 acc[8], ve[8];
    const int r_cur = st.r_pref;
    int rb = st.rb;
#pragma unroll
    for (int mb = 0; mb < 8; ++mb) {
      ve[mb] = st.ve[mb];
      acc[mb] = lb_pk_add(st.ps[mb], st.pr[mb]);
    }
    asm volatile("" : "+v"(rb));
    if (first) lb_range_probe(a.ctrl, ve, 8);
    if constexpr (PRIO) __builtin_amdgcn_s_setprio(2);
    if constexpr (!(ABL & 8)) lb_gemm16v<false>(w0b, ve, acc);
    if (first) lb_range_probe(a.ctrl, acc, 8);
    f32x4 acc2[8];
#pragma unroll
    for (int mb = 0; mb < 8; ++mb) acc2[mb] = vecb[4 * mb];
    if constexpr (!(ABL & 8)) {
      lb_gemm16v<true>(w1b, acc, acc2);
    } else {
#pragma unroll
      for (int mb = 0; mb < 8; ++mb)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc2[mb][j] += acc[mb][j] + ve[mb][j];
    }
    if constexpr (PRIO) __builtin_amdgcn_s_setprio(0);
    // ---- late prefetch: the next tile's loads and the indices of the one after
    {
      int s = st.s_n, r = st.r_n;
      asm volatile("" : "+v"(s), "+v"(r));
      issue(min(tc + stride, t_last), s, r);
      const int64_t rn = rowc_of(min(tc + 2 * stride, t_last));
      st.s_n = a.senders[rn];
      st.r_n = a.receivers[rn];
    }
    f32x4 y[8];
    lb_layernorm16(acc2, vecb + 32, vecb + 64, y, ln_inv_d, ln_pad);
    const int row = tc * 16 + n;
    const bool valid = row < E;
    if constexpr (!SKIP && !(ABL & 4)) {
      f32x4* ew = reinterpret_cast<f32x4*>(a.elat_out ? a.elat_out : a.elat) + (int64_t)tc * 512 + lane;
#pragma unroll
      for (int mb = 0; mb < 8; ++mb) __builtin_nontemporal_store(lb_pk_add(ve[mb], y[mb]), &ew[64 * mb]);
    }
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
#pragma unroll
    for (int mb = 0; mb < 8; mb += 2) lb_scan8(y[mb], y[mb + 1], m1, m2, m4, m8);
    if (tail && valid && !(ABL & 4)) {
      int slot01;
      const bool complete = lb_seg_complete(rb, rr, segstart, n, tc, E, slot01);
      float* dst = complete ? a.agg + (int64_t)rr * 128 : a.part + ((int64_t)tc * 2 + slot01) * 128;
      f32x4* d4 = reinterpret_cast<f32x4*>(dst) + g;
#pragma unroll
      for (int mb = 0; mb < 8; ++mb) d4[4 * mb] = y[mb];
    }
  };
  body(t, true);  // peeled (see the header comment)
  t += stride;
  for (int it = 1; it < n_iter; ++it, t += stride) body(t, false);
}

// ---------------------------------------------------------------------------------------------
// k_edge16p: the same tile body, fully software-pipelined inside each wave.  While tile t is in the
// GEMMs, ALL loads of tile t+1 (edge latents + gathered sender/receiver projections + CSR bounds) and
// the indices of tile t+2 are in flight, so the bytes a CU has outstanding no longer depend on how
// the phases of its waves happen to line up (fine-grained arbitration keeps equal waves in lock step:
// they all load, then all compute).  The prefetch sets cost 96 VGPRs -> WPS = 2 (256 VGPRs).
// The edge latents are touched once per layer (562 MB >> L2 + Infinity Cache): their loads and stores
// are nontemporal so that the streams do not push the gathered psr table out of L2 / MALL
// (tools/stream_bench: 257 -> 233 us for this access pattern).
// The first tile is peeled: a waitcnt at the loop header serves the entry and the back edge with ONE
// static count, and the entry's smaller count would drain the previous tile's stores on every trip.
struct lb_e16p_state {
  f32x4 ve[8], ps[8], pr[8];  // next tile, in flight
  int s_n, r_n;               // indices of the tile after next
  int r_pref, rb;             // receiver of the next tile's lanes + its boundary probe
};

template <int WPS, bool SKIP, int ABL = 0>
__global__ void __launch_bounds__(WPS * 256, WPS) k_edge16p(lb_edge16_args a) {
  constexpr int THREADS = WPS * 256, WAVES = WPS * 4;
  constexpr int NW0 = 4096;
  __shared__ f32x4 sW[NW0 + 4096 + 96];
  if (a.ctrl->overflow_step >= 0) return;
  const int tid = threadIdx.x;
  {
    const f32x4* g0 = reinterpret_cast<const f32x4*>(a.w0p);
    const f32x4* g1 = reinterpret_cast<const f32x4*>(a.w1p);
    for (int i = tid; i < NW0; i += THREADS) sW[i] = g0[i];
    for (int i = tid; i < 4096; i += THREADS) sW[NW0 + i] = g1[i];
    if (tid < 96) {
      const float* src = tid < 32 ? a.b1 : (tid < 64 ? a.ln_s : a.ln_o);
      sW[NW0 + 4096 + tid] = reinterpret_cast<const f32x4*>(src)[tid & 31];
    }
  }
  __syncthreads();
  const int E = a.ctrl->n_edges_total;
  const float ln_inv_d = a.ctrl->ln_inv_d, ln_pad = a.ctrl->ln_pad;
  const int ntiles = (E + 15) >> 4;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, g = lane >> 4;
  const int xcd = blockIdx.x & 7, slot = (blockIdx.x >> 3) * WAVES + wave;
  const int stride = (gridDim.x >> 3) * WAVES;
  const int t_lo = (int)(((int64_t)ntiles * xcd) >> 3), t_hi = (int)(((int64_t)ntiles * (xcd + 1)) >> 3);
  int t = t_lo + slot;
  if (t >= t_hi) return;
  uint32_t off0 = (uint32_t)(uintptr_t)(lds_cptr)(sW + lane);
  uint32_t off1 = (uint32_t)(uintptr_t)(lds_cptr)(sW + NW0 + lane);
  uint32_t off2 = (uint32_t)(uintptr_t)(lds_cptr)(sW + NW0 + 4096 + g);
  asm volatile("" : "+v"(off0), "+v"(off1), "+v"(off2));
  const lds_cptr w0b = (lds_cptr)(uintptr_t)off0, w1b = (lds_cptr)(uintptr_t)off1, vecb = (lds_cptr)(uintptr_t)off2;
  auto rowc_of = [&](int tt) -> int64_t {
    const int row = tt * 16 + n;
    return row < E ? row : E - 1;
  };
  const f32x4* psr4 = reinterpret_cast<const f32x4*>(a.psr);
  const int n_iter = (t_hi - 1 - t) / stride + 1;
  const int t_last = t + (n_iter - 1) * stride;

  lb_e16p_state st;
  // every load is unconditional (tile indices clamp to the wave's last tile): a load under a
  // branch makes the compiler wait vmcnt(0) for the loop-carried registers
  auto issue = [&](int tt, int s, int r) {
    const f32x4* er = reinterpret_cast<const f32x4*>(a.elat) + (int64_t)tt * 512 + lane;
    const f32x4* ps = psr4 + (int64_t)s * 64 + g;
    const f32x4* pr = psr4 + (int64_t)r * 64 + 32 + g;
#pragma unroll
    for (int mb = 0; mb < 8; ++mb) {
      st.ve[mb] = (ABL & 2) ? f32x4{1.f, 2.f, (float)tt, (float)mb} : __builtin_nontemporal_load(&er[64 * mb]);
      st.ps[mb] = (ABL & 1) ? f32x4{.1f, .2f, (float)s, (float)mb} : ps[4 * mb];
      st.pr[mb] = (ABL & 1) ? f32x4{.3f, .1f, (float)r, (float)mb} : pr[4 * mb];
    }
    st.r_pref = r;
    st.rb = lb_edge_probe(a.receivers, tt, lane, E);
  };
  {
    const int64_t rc = rowc_of(t);
    int s0 = a.senders[rc], r0 = a.receivers[rc];
    asm volatile("" : "+v"(s0), "+v"(r0));
    issue(t, s0, r0);
    const int64_t rn = rowc_of(min(t + stride, t_last));
    st.s_n = a.senders[rn];
    st.r_n = a.receivers[rn];
  }

  auto body = [&](int tc, bool first) {
    // ---- take delivery of the prefetched tile
    f32x4 acc[8], ve[8];
    const int r_cur = st.r_pref;
    int rb = st.rb;
#pragma unroll
    for (int mb = 0; mb < 8; ++mb) {
      ve[mb] = st.ve[mb];
      acc[mb] = lb_pk_add(st.ps[mb], st.pr[mb]);
    }
    asm volatile("" : "+v"(rb));
    // ---- put the next tile's loads and the indices of the one after in flight
    {
      int s = st.s_n, r = st.r_n;
      asm volatile("" : "+v"(s), "+v"(r));
      issue(min(tc + stride, t_last), s, r);
      const int64_t rn = rowc_of(min(tc + 2 * stride, t_last));
      st.s_n = a.senders[rn];
      st.r_n = a.receivers[rn];
    }
    if (first) lb_range_probe(a.ctrl, ve, 8);
    if constexpr (!(ABL & 8)) lb_gemm16v<false>(w0b, ve, acc);
    if (first) lb_range_probe(a.ctrl, acc, 8);
    f32x4 acc2[8];
#pragma unroll
    for (int mb = 0; mb < 8; ++mb) acc2[mb] = vecb[4 * mb];
    if constexpr (!(ABL & 8)) {
      lb_gemm16v<true>(w1b, acc, acc2);
    } else {
#pragma unroll
      for (int mb = 0; mb < 8; ++mb)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc2[mb][j] += acc[mb][j] + ve[mb][j];
    }
    f32x4 y[8];
    lb_layernorm16(acc2, vecb + 32, vecb + 64, y, ln_inv_d, ln_pad);
    const int row = tc * 16 + n;
    const bool valid = row < E;
    if constexpr (!SKIP && !(ABL & 4)) {
      f32x4* ew = reinterpret_cast<f32x4*>(a.elat_out ? a.elat_out : a.elat) + (int64_t)tc * 512 + lane;
#pragma unroll
      for (int mb = 0; mb < 8; ++mb) __builtin_nontemporal_store(lb_pk_add(ve[mb], y[mb]), &ew[64 * mb]);
    }
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
#pragma unroll
    for (int mb = 0; mb < 8; mb += 2) lb_scan8(y[mb], y[mb + 1], m1, m2, m4, m8);
    if (tail && valid && !(ABL & 4)) {
      int slot01;
      const bool complete = lb_seg_complete(rb, rr, segstart, n, tc, E, slot01);
      float* dst = complete ? a.agg + (int64_t)rr * 128 : a.part + ((int64_t)tc * 2 + slot01) * 128;
      f32x4* d4 = reinterpret_cast<f32x4*>(dst) + g;
#pragma unroll
      for (int mb = 0; mb < 8; ++mb) d4[4 * mb] = y[mb];
    }
  };
  body(t, true);  // peeled (see the header comment)
  t += stride;
  for (int it = 1; it < n_iter; ++it, t += stride) body(t, false);
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
  static const int wps_env = getenv("LB_ENC_WPS") ? atoi(getenv("LB_ENC_WPS")) : 4;
  if (tiles_cap <= 256 * 8 * 2)
    hipLaunchKernelGGL((k_edge_enc16v<2>), dim3(grid_for(8)), dim3(512), 0, e->stream, a);
  else if (wps_env == 2)
    hipLaunchKernelGGL((k_edge_enc16v<2>), dim3(256), dim3(512), 0, e->stream, a);
  else if (wps_env == 3)
    hipLaunchKernelGGL((k_edge_enc16v<3>), dim3(256), dim3(768), 0, e->stream, a);
  else
    hipLaunchKernelGGL((k_edge_enc16v<4>), dim3(256), dim3(1024), 0, e->stream, a);
  LB_HIP(hipGetLastError());
  return LB_OK;
}

int lbk_edge16v(lb_engine* e, const lb_edge16_args& a, int variant) {
#define LB_E16V(W, R, G)                                                                                \
  do {                                                                                                  \
    if (a.skip_elat_store)                                                                              \
      LB_LAUNCH_TIMED(e, (k_edge16v<W, R, true, 0, true>), dim3(G), dim3(W * 256), a);                  \
    else                                                                                                \
      LB_LAUNCH_TIMED(e, (k_edge16v<W, R, false, 0, true>), dim3(G), dim3(W * 256), a);                 \
  } while (0)
  // Small graphs (one 2.5 k-particle trajectory = ~1000 tiles): a launch is the latency chain
  // "stage 133 KiB of weights -> one tile per wave", so launch no more workgroups than there are tiles for.
  // The tile count is bounded on the host by the frozen capacity (the real count lives on the device).
  const int64_t tiles_cap = ((int64_t)e->e_cap * e->g.B + 15) / 16;
  auto grid_for = [&](int waves_per_block) {
    int64_t g = (tiles_cap + waves_per_block - 1) / waves_per_block;
    g = (g + 7) / 8 * 8;  // the XCD-aware walk wants a multiple of 8
    return (int)(g < 8 ? 8 : (g > 256 ? 256 : g));
  };
  if (variant == 0 && tiles_cap <= 256 * 8 * 2) {
    // (one wave per SIMD, i.e. twice the workgroups, measures slower: 17 vs 13.8 us per launch on a 20 k-edge
    // graph - every workgroup stages the 133 KiB of weights; three waves per SIMD: 15.4 us)
    LB_E16V(2, false, grid_for(8));
  } else {
    switch (variant) {
      case 0: LB_E16V(2, false, 256); break;   // default: two waves per SIMD measure faster than three
      case 6: LB_E16V(3, false, 256); break;
      case 1: LB_E16V(4, true, 256); break;
      case 2: LB_E16V(3, true, 256); break;
      case 4:
        if (a.skip_elat_store)
          hipLaunchKernelGGL((k_edge16l<2, true>), dim3(256), dim3(512), 0, e->stream, a);
        else
          hipLaunchKernelGGL((k_edge16l<2, false>), dim3(256), dim3(512), 0, e->stream, a);
        break;
      case 5: LB_E16V(2, false, 256); break;
      case 3:
        if (a.skip_elat_store)
          hipLaunchKernelGGL((k_edge16p<2, true>), dim3(256), dim3(512), 0, e->stream, a);
        else
          hipLaunchKernelGGL((k_edge16p<2, false>), dim3(256), dim3(512), 0, e->stream, a);
        break;
      default: return lb_fail(LB_ERR_ARG, "k_edge16v variant %d", variant);
    }
  }
#undef LB_E16V
  LB_HIP(hipGetLastError());
  return LB_OK;
}
