// lb_segnn_msg.hip - the SEGNN message and update functions of one layer, ONE kernel each:
//   k_sg_msg:  gather f_sender, f_receiver -> O3TensorProductGate -> O3TensorProductGate -> segment_sum
//   k_sg_upd:  [f | agg] -> O3TensorProductGate -> O3TensorProduct -> f +=
// (SEGNNLayer._message / _update + jraph aggregation, lagrangebench/models/segnn.py:280-334; e3nn
// conventions as in oracle/segnn_oracle.py).  blocks_per_step == 2 only; other depths use the
// per-block kernel of lb_segnn.hip.
//
// Why one kernel: SEGNN keeps no edge state between layers, so with the two blocks chained in
// registers and the aggregation fused into the epilogue NOTHING edge-sized is written: per edge the
// kernel reads two 512-B node rows (L2/MALL resident) + 40 B of list data, and per receiver it
// writes one 512-B row.
//
// Arithmetic: 16-edge tiles on v_mfma_f32_16x16x32_f16 in the f16x2 split scheme of lb_f16x2.h
// (x = hi + lo, products lo*hi + hi*lo + hi*hi, fp32 accumulate).  Layout: lane (n = l&15, g = l>>4)
// holds, for edge n, the features 16 mb + 4 g + j of a 128-float SV row [s | vx | vy | vz]
// (mb 0,1 = scalars, 2,3 = x, 4,5 = y, 6,7 = z); a K-step of 32 is exactly one of those groups.
//
// Algebra used to cut work (a0 = Y0 is the same for every edge, a = Y1 r/|r|):
//   out_s    = b + [s | sum_c a_c v_c] [Y0 Ws_s ; Ws_v/sqrt3]          one modulated operand only
//   out_v[c] = a_c (s Wv_s) + v_c (Y0 Wv_v)                            s Wv_s computed once, not 3x
// (constants folded into the weights on the host); the two message features (|r|, r) enter through
// fp32 FMAs on the accumulator start values.  Gate activations and the gating product are in-lane:
// the gate of vector channel m sits in the same (lane, register) as the channel itself.
//
// Round 4 (the kernel is bound by VALU issue, not by the matrix pipe or by memory: 916 VALU + 66
// transcendental + 144 MFMA instructions per tile in round 3):
//   * 2D cases never touch the z component: a_z = 0 and v_z = 0 on every node, so out_v[z] is exactly 0
//     in the reference too - the DIM = 2 instantiation skips its gathers, splits, MFMAs, gates, scan;
//   * the gate constants live in the weights: every S column is pre-scaled by -log2(e) (sigmoid(x) =
//     rcp(1 + exp2(z)), silu(x) ~ z * sigmoid: v_exp_f32 + v_add + v_rcp_f32 [+ v_mul]), the
//     second-moment constant of the sigmoid sits in the T / V matrices, the one of the silu in the rows
//     of the NEXT block that consume it;
//   * the message feature r enters through T (r_c = a_c |r| / Y1: one FMA per T entry instead of one
//     multiply per V entry);
//   * fp16 hi/lo split on the mixed-precision fma (lb_split8v: 12 instead of 20 VALU per 8 values), the
//     segmented scan as v_fmac_f32_dpp asm blocks (lb_scan8: no v_mov_dpp + packed-fma pairs), no
//     row_ptr gathers (lb_edge_probe), no zeroing of invalid rows (they are their own segments and
//     are never stored);
//   * per k-group ONE explicit pair of s_waitcnt (lo fragments, then hi fragments) and s_setprio around
//     the MFMA phases, as in lb_gemm16v;
//   * prologue: the weight image travels through registers and the first tile's indices are requested
//     while it is in flight (one dependent round trip less - the B = 1 launches are latency chains).
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "lb_segnn_dev.h"

// LDS image of one message layer, in f32x4 units (lb_sg_msg_image below)
#define SGM_WS0 0      // K=128 x M=64  hi|lo: 4 p x 4 mbo x 2 x 64
#define SGM_WT0 2048   // K=64  x M=32: 2 x 2 x 2 x 64
#define SGM_WV0 2560
#define SGM_B1 3072    // second lane base (the ds offset field is 16 bits)
#define SGM_WS1 3072   // K=64 x M=64: 2 x 4 x 2 x 64
#define SGM_WT1 4096   // K=32 x M=32: 1 x 2 x 2 x 64
#define SGM_WV1 4352
#define SGM_VEC 4608   // fp32 vectors: b0(16) wdS(16) wrS(16) wdT(8) wrT(8) b1(16) = 80 f32x4
#define SGM_IMAGE 4688

struct lb_sg_msg_args {
  const lb_ctrl* ctrl;
  const int32_t* senders;
  const int32_t* receivers;
  const float* efeat;   // [E][8]
  const float* f;       // [BN][128] hidden state
  const float* image;   // SGM_IMAGE f32x4 of this layer
  float* agg;           // [BN][128]
  float* part;          // [ceil(E/16)][2][128]; lives in the SAME allocation as agg (one buffer descriptor for the stores)
  uint32_t part_off;    // byte offset of part from agg
  uint32_t out_bytes;   // bytes of the agg | part allocation (< 2^31)
  long long* dbg;       // ABL & 32 (round 4's tools/sg_msg_bench, commit 9d3d71e): per wave of workgroup 0, cycles per tile segment
};

// WPS waves per SIMD (one workgroup of WPS * 256 threads per CU); each wave loads its own tile (only the
// next tile's two indices travel ahead) and the other waves of the SIMD hide the latency.
// ABL (round 4's tools/sg_msg_bench.hip - removed with the museum kernel it cross-checked, commit 9d3d71e has it; 0 in the product): 1 no row gathers, 2 no MFMAs, 4 no gates, 8 no scan, 16 no stores.
template <int DIM, int WPS, bool PRIO, int ABL = 0, int SM = 0x1ff>
__global__ void __launch_bounds__(WPS * 256, 1) k_sg_msg(lb_sg_msg_args a) {
  constexpr int NT = WPS * 256, WAVES = WPS * 4, NC = DIM;
  __shared__ f32x4 sW[SGM_IMAGE];
  __shared__ int s_ticket;
  // prologue order as in k_edge16v: control block, weight loads into registers, the first tile's indices
  // while those are in flight, poison check, LDS writes, barrier
  const int poisoned = a.ctrl->overflow_step;
  const int E = a.ctrl->n_edges_total;
  const int tid = threadIdx.x;
  constexpr int NST = (SGM_IMAGE + NT - 1) / NT;
  f32x4 st[NST];
  {
    const f32x4* src = reinterpret_cast<const f32x4*>(a.image);
#pragma unroll
    for (int k = 0; k < NST; ++k) {
      const int i = tid + k * NT;
      st[k] = src[i < SGM_IMAGE ? i : SGM_IMAGE - 1];
    }
  }
  const int ntiles = (E + 15) >> 4;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, g = lane >> 4;
  // Tile walk: XCD x (= blockIdx % 8) owns a contiguous eighth of the receiver-sorted list, each of its workgroups a
  // contiguous chunk of that, and the waves of a workgroup draw tiles from an LDS ticket counter.  (A static strided
  // walk lets the SIMD's oldest wave - the issue arbiter prefers it - finish its tiles at ~55 % of the launch and
  // leaves the youngest to run alone at the end: stamps in profiles/r04_sg_msg_bench.txt.)  Which wave computes a
  // tile does not change a bit of the result: aggregates and partial slots are per tile.
  const int xcd = blockIdx.x & 7, wg = blockIdx.x >> 3, nwg = gridDim.x >> 3;
  const int t_lo = (int)(((int64_t)ntiles * xcd) >> 3), t_hi = (int)(((int64_t)ntiles * (xcd + 1)) >> 3);
  const int per = (t_hi - t_lo + nwg - 1) / nwg;
  const int c_lo = t_lo + wg * per, c_hi = min(t_hi, c_lo + per);
  int t = c_lo + wave;  // first tile: static, so that its indices can be requested before the barrier
  auto rowc_of = [&](int tt) -> int64_t {
    const int row = tt * 16 + n;
    return row < E ? row : (E > 0 ? E - 1 : 0);
  };
  int s_c = 0, r_c = 0;
  if (t < c_hi) {
    const int64_t rc = rowc_of(t);
    s_c = a.senders[rc];
    r_c = a.receivers[rc];
  }
  if (poisoned >= 0) return;
#pragma unroll
  for (int k = 0; k < NST; ++k) {
    const int i = tid + k * NT;
    if (i < SGM_IMAGE) sW[i] = st[k];
  }
  if (tid == 0) s_ticket = WAVES;
  __syncthreads();
  if (t >= c_hi) return;
  uint32_t off0 = (uint32_t)(uintptr_t)(lds_cptr)(sW + lane);
  uint32_t off1 = (uint32_t)(uintptr_t)(lds_cptr)(sW + SGM_B1 + lane);
  uint32_t off2 = (uint32_t)(uintptr_t)(lds_cptr)(sW + SGM_VEC + g);
  asm volatile("" : "+v"(off0), "+v"(off1), "+v"(off2));
  const lds_cptr w0 = (lds_cptr)(uintptr_t)off0, w1 = (lds_cptr)(uintptr_t)off1, vec = (lds_cptr)(uintptr_t)off2;
  const f32x4* ef4 = reinterpret_cast<const f32x4*>(a.efeat);
  int n_iter = 0;
  asm volatile("" : "+v"(s_c), "+v"(r_c));
  // Memory schedule of a tile (gfx9 has ONE in-order vmcnt for loads and stores): the SENDER rows of tile t + 1 are
  // requested in the middle of tile t (their registers are free once block 0 is done), i.e. BEFORE tile t's aggregate
  // stores; the receiver rows, the edge features and the indices of the tile after go out at the top of the tile.  The
  // stores are raw-buffer stores with the lane's offset pushed out of range where it has nothing to write: no branch
  // around them, so the compiler counts them exactly and the wait for the sender rows (older than the stores) does not
  // drain the stores (behind a branch every later wait became vmcnt(0): ~5 us of a 50 us launch, ablation "no stores").
  const __amdgpu_buffer_rsrc_t out_rs = __builtin_amdgcn_make_buffer_rsrc(a.agg, 0, (int)a.out_bytes, 0x00020000);
  f32x4 fs[8];
  auto issue_sender = [&](int s_idx) {
    const f32x4* ps = reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(a.f) + ((uint32_t)s_idx * 512u + (uint32_t)g * 16u));
#pragma unroll
    for (int mb = 0; mb < 2 + 2 * NC; ++mb) fs[mb] = (ABL & 1) ? f32x4{.1f, .2f, (float)s_idx * 1e-6f, (float)mb} : ps[4 * mb];
  };
  issue_sender(s_c);
  f32x4 ef_n = ef4[rowc_of(t) * 2];  // edge features travel with the sender rows (the accumulator start values need them first)
  // (the first tile's rows are taken delivery of HERE: a wait at the loop header is also executed on the back edge, where
  // the waitcnt pass would have to assume the preheader's state - nothing younger in flight - and drain the stores)
  asm volatile("" : "+v"(ef_n), "+v"(fs[0]), "+v"(fs[1]), "+v"(fs[2]), "+v"(fs[3]), "+v"(fs[4]), "+v"(fs[5]));
  if constexpr (NC == 3) asm volatile("" : "+v"(fs[6]), "+v"(fs[7]));
  long long stamp[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tprev = 0;
#define SG_STAMP(i)                                         \
  if constexpr ((ABL & 32) && ((SM >> i) & 1)) {            \
    const long long now = __builtin_readcyclecounter();     \
    stamp[i] += now - tprev;                                \
    tprev = now;                                            \
  }                                                         \
  if constexpr (ABL & 64) __builtin_amdgcn_s_sleep(1);      \
  if constexpr (ABL & 128) {                                \
    SB();                                                   \
    __builtin_amdgcn_s_waitcnt(LB_WAIT_LGKM(0));            \
    SB();                                                   \
  }                                                         \
  if constexpr (ABL & 256) {                                \
    (void)__builtin_amdgcn_s_memtime();                     \
  }
  if constexpr (ABL & 32) tprev = __builtin_readcyclecounter();

  while (t < c_hi) {
    ++n_iter;
    // the next tile's ticket (one lane draws, the wave reads it back as a scalar)
    int t_next;
    {
      int k = 0;
      if (lane == 0) k = atomicAdd(&s_ticket, 1);
      t_next = c_lo + __builtin_amdgcn_readfirstlane(k);
    }
    const int r_cur = r_c;
    f32x4 fr[8];
    {
      // 32-bit byte offsets from the scalar base (global_load saddr form: no 64-bit address arithmetic per lane;
      // the host refuses node tables beyond 4 GiB)
      const f32x4* pr = reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(a.f) + ((uint32_t)r_c * 512u + (uint32_t)g * 16u));
#pragma unroll
      for (int mb = 0; mb < 2 + 2 * NC; ++mb) fr[mb] = (ABL & 1) ? f32x4{.3f, .1f, (float)r_c * 1e-6f, (float)mb} : pr[4 * mb];
    }
    f32x4 ef = ef_n;
    {
      const int64_t rn = rowc_of(min(t_next, c_hi - 1));
      s_c = a.senders[rn];
      r_c = a.receivers[rn];
    }
    int rb = lb_edge_probe(a.receivers, t, lane, E);
    // edge attribute a = Y1 r/|r| (0 for the self edge), message features |r| (rel_dist) and r
    float at[3], dist, dotr, rmag;
    {
      const float rx = ef[0], ry = ef[1], rz = DIM == 3 ? ef[2] : 0.f;
      dist = DIM == 3 ? ef[3] : ef[2];
      const float n2 = DIM == 3 ? rx * rx + ry * ry + rz * rz : rx * rx + ry * ry;
      const float rs = n2 == 0.f ? 0.f : __builtin_amdgcn_rsqf(n2);  // v_rsq_f32 (1 ulp); 0 for the self edge
      const float inv = SG_Y1 * rs;
      at[0] = rx * inv;
      at[1] = ry * inv;
      at[2] = DIM == 3 ? rz * inv : 0.f;
      dotr = (n2 * rs) * SG_Y1;            // r . a = |r| Y1
      rmag = (n2 * rs) * (1.0f / SG_Y1);   // r_c = a_c * rmag
    }
    SG_STAMP(0)  // loads issued, attribute arithmetic
    // ---- block 0: [f_s | f_r | (r, |r|)] (x) a -> gate
    f32x4 S[4], T[2], V[3][2];
#pragma unroll
    for (int m = 0; m < 4; ++m) S[m] = vec[4 * m] + vec[16 + 4 * m] * dist + vec[32 + 4 * m] * dotr;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      T[m] = vec[48 + 4 * m] * dist + vec[56 + 4 * m] * rmag;
#pragma unroll
      for (int c = 0; c < 3; ++c) V[c][m] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    SG_STAMP(1)  // accumulator start values (LDS vectors)
    sg_operand<NC, 4, 4, PRIO, (ABL & 2) != 0>(w0 + SGM_WS0 * 1, w0 + SGM_WT0, w0 + SGM_WV0, fs, at, S, T, V);
    SG_STAMP(2)  // sender operand (incl. the wait for its rows)
    sg_operand<NC, 4, 4, PRIO, (ABL & 2) != 0>(w0 + SGM_WS0 + 2 * 512, w0 + SGM_WT0 + 256, w0 + SGM_WV0 + 256, fr, at, S, T, V);
    SG_STAMP(3)  // receiver operand
    f32x4 H[8];
    if constexpr (ABL & 4) {
#pragma unroll
      for (int mb = 0; mb < 2 + 2 * NC; ++mb) H[mb] = mb < 4 ? S[mb] : (mb < 6 ? T[mb - 4] + V[0][mb - 4] : V[1][mb - 6]);
    } else {
      sg_gate<NC>(S, T, V, at, H);
    }
    // the next tile's sender rows, into the registers block 0 has just released (see the memory schedule above)
    asm volatile("" : "+v"(s_c), "+v"(r_c), "+v"(rb));
    issue_sender(s_c);
    ef_n = ef4[rowc_of(min(t_next, c_hi - 1)) * 2];
    SB();
    SG_STAMP(4)  // gate 0
    // ---- block 1: h (x) a -> gate
#pragma unroll
    for (int m = 0; m < 4; ++m) S[m] = vec[64 + 4 * m];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      T[m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < 3; ++c) V[c][m] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    sg_operand<NC, 4, 4, PRIO, (ABL & 2) != 0>(w1 + (SGM_WS1 - SGM_B1), w1 + (SGM_WT1 - SGM_B1), w1 + (SGM_WV1 - SGM_B1), H, at, S, T, V);
    SG_STAMP(5)  // block 1 operand
    f32x4 y[8];
    if constexpr (ABL & 4) {
#pragma unroll
      for (int mb = 0; mb < 2 + 2 * NC; ++mb) y[mb] = mb < 4 ? S[mb] : (mb < 6 ? T[mb - 4] + V[0][mb - 4] : V[1][mb - 6]);
    } else {
      sg_gate<NC>(S, T, V, at, y);
    }
    y[0] = y[0] * SG_K_SILU;
    y[1] = y[1] * SG_K_SILU;
    SG_STAMP(6)  // gate 1
    // ---- fused segment_sum over the receiver-sorted list (k_edge16v's epilogue)
    const int row = t * 16 + n;
    const bool valid = row < E;
    const int rr = valid ? r_cur : (-1 - n);
    const int r_prev = __builtin_amdgcn_update_dpp(-2, rr, 0x111, 0xF, 0xF, false);
    const bool head = (n == 0) || (rr != r_prev);
    const unsigned Hm = (unsigned)(__ballot(head) & 0xffffull);
    const unsigned below = Hm & ((2u << n) - 1u);
    const int segstart = 31 - __clz(below);
    const bool tail = (n == 15) || ((Hm >> (n + 1)) & 1u);
    const float m1 = (n >= 1 && segstart <= n - 1) ? 1.f : 0.f, m2 = (n >= 2 && segstart <= n - 2) ? 1.f : 0.f;
    const float m4 = (n >= 4 && segstart <= n - 4) ? 1.f : 0.f, m8 = (n >= 8 && segstart <= n - 8) ? 1.f : 0.f;
    // (rows past the end of the list are copies of the last edge: finite, each its own segment, never stored)
    if constexpr (!(ABL & 8))
#pragma unroll
    for (int mb = 0; mb < 2 + 2 * NC; mb += 2) lb_scan8(y[mb], y[mb + 1], m1, m2, m4, m8);
    SG_STAMP(7)  // scan
    if constexpr (!(ABL & 16)) {
      int slot01;
      const bool complete = lb_seg_complete(rb, rr, segstart, n, t, E, slot01);
      const uint32_t row_off = complete ? (uint32_t)rr * 512u : a.part_off + ((uint32_t)t * 2u + (uint32_t)slot01) * 512u;
      const uint32_t off = (tail && valid) ? row_off + (uint32_t)g * 16u : 0x80000000u;  // out of range: dropped
      typedef uint32_t u32x4b __attribute__((ext_vector_type(4)));
#pragma unroll
      for (int mb = 0; mb < 8; ++mb) {
        const f32x4 v = mb < 2 + 2 * NC ? y[mb] : f32x4{0.f, 0.f, 0.f, 0.f};  // 2D: consumers outside this file read whole rows
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4b, v), out_rs, (int)off, 64 * mb, 0);
      }
    }
    SG_STAMP(8)  // stores issued
    t = t_next;
  }
  if constexpr (ABL & 32) {
    if (blockIdx.x == 0 && lane == 0) {
#pragma unroll
      for (int i = 0; i < 9; ++i) a.dbg[wave * 10 + i] = stamp[i];
      a.dbg[wave * 10 + 9] = n_iter;
    }
  }
#undef SG_STAMP
}

// Rows cut by a tile boundary: sum their per-tile partial slots in tile order (deterministic);
// rows without edges get zeros.  32 lanes per node, lane c owns the 16-byte chunk c.
__global__ void __launch_bounds__(256) k_sg_agg_finish(const lb_ctrl* __restrict__ ctrl, int64_t n_rows,
                                                       const int32_t* __restrict__ row_ptr,
                                                       const float* __restrict__ part,
                                                       float* __restrict__ agg) {
  if (ctrl->overflow_step >= 0) return;
  const int64_t node = (int64_t)blockIdx.x * 8 + (threadIdx.x >> 5);
  const int c = threadIdx.x & 31;
  if (node >= n_rows) return;
  const int E = ctrl->n_edges_total;
  int k0 = row_ptr[node], k1 = row_ptr[node + 1];
  k0 = k0 < E ? k0 : E;
  k1 = k1 < E ? k1 : E;
  f32x4* out = reinterpret_cast<f32x4*>(agg) + node * 32 + c;
  if (k1 <= k0) {
    *out = f32x4{0.f, 0.f, 0.f, 0.f};
    return;
  }
  const int t0 = k0 >> 4, t1 = (k1 - 1) >> 4;
  if (t0 == t1) return;  // written whole by k_sg_msg
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  for (int t = t0; t <= t1; ++t)
    s = s + reinterpret_cast<const f32x4*>(part + ((int64_t)t * 2 + (k0 <= (t << 4) ? 0 : 1)) * 128)[c];
  *out = s;
}

// ------------------------------------------------------------------------------ node update
// SEGNNLayer._update (segnn.py:306-334) for blocks_per_step == 2 as one kernel:
//   [f | agg] -> O3TensorProductGate -> O3TensorProduct -> f += .   Node attribute a0 == 1.
// Same register-chained f16x2 scheme as k_sg_msg, on tiles of 16 consecutive nodes.
#define SGU_WS0 0      // K=128 x M=64
#define SGU_WT0 2048   // K=64 x M=32
#define SGU_WV0 2560
#define SGU_WS1 3072   // K=64 x M=32 (the last block has no gates): 2 x 2 x 2 x 64
#define SGU_WT1 3584
#define SGU_WV1 3840
#define SGU_VEC 4096   // b0 (16 f32x4), b1 (8)
#define SGU_IMAGE 4120

struct lb_sg_upd_args {
  const lb_ctrl* ctrl;
  int64_t n_rows;
  float* f;             // [rows][128] in/out
  const float* agg;     // [rows][128]
  const float* nattr;   // [rows][4]
  const float* image;
  const int32_t* row_ptr;  // with part != null: combine k_sg_msg's partial slots here
  const float* part;
};

template <int DIM, int NT>
__global__ void __launch_bounds__(NT, 2) k_sg_upd(lb_sg_upd_args a) {
  constexpr int NC = DIM, WAVES = NT / 64, NMB = 2 + 2 * NC;
  __shared__ f32x4 sW[SGU_IMAGE];
  const int poisoned = a.ctrl->overflow_step;  // acted on after the staging loads are in flight, see k_sg_msg
  const int tid = threadIdx.x;
  constexpr int NST = (SGU_IMAGE + NT - 1) / NT;
  f32x4 st[NST];
  {
    const f32x4* src = reinterpret_cast<const f32x4*>(a.image);
#pragma unroll
    for (int k = 0; k < NST; ++k) {
      const int i = tid + k * NT;
      st[k] = src[i < SGU_IMAGE ? i : SGU_IMAGE - 1];
    }
  }
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, g = lane >> 4;
  const int ntiles = (int)((a.n_rows + 15) >> 4);
  const int E = a.ctrl->n_edges_total;
  int t = blockIdx.x * WAVES + wave;
  // the first tile's row_ptr pair travels with the weights
  int k0n = 0, k1n = 0;
  if (a.part != nullptr && t < ntiles) {
    const int64_t row = (int64_t)t * 16 + n;
    const int64_t rl = row < a.n_rows ? row : a.n_rows - 1;
    k0n = a.row_ptr[rl];
    k1n = a.row_ptr[rl + 1];
  }
  if (poisoned >= 0) return;
#pragma unroll
  for (int k = 0; k < NST; ++k) {
    const int i = tid + k * NT;
    if (i < SGU_IMAGE) sW[i] = st[k];
  }
  __syncthreads();
  const lds_cptr w0 = (lds_cptr)(sW + lane), vec = (lds_cptr)(sW + SGU_VEC + g);
  for (; t < ntiles; t += gridDim.x * WAVES) {
    const int64_t row = (int64_t)t * 16 + n;
    const bool valid = row < a.n_rows;
    const int64_t rl = valid ? row : a.n_rows - 1;
    f32x4* frow = reinterpret_cast<f32x4*>(a.f) + rl * 32 + g;
    const f32x4* arow = reinterpret_cast<const f32x4*>(a.agg) + rl * 32 + g;
    f32x4 X0[8], X1[8];
#pragma unroll
    for (int mb = 0; mb < NMB; ++mb) X0[mb] = frow[4 * mb];
    if (a.part == nullptr) {
#pragma unroll
      for (int mb = 0; mb < NMB; ++mb) X1[mb] = arow[4 * mb];
    } else {
      // aggregated messages straight from k_sg_msg: whole rows sit in agg, rows cut by a 16-edge
      // tile boundary are the sum of their per-tile partial slots in tile order (deterministic)
      int k0 = k0n, k1 = k1n;
      k0 = k0 < E ? k0 : E;
      k1 = k1 < E ? k1 : E;
      {  // next tile's pair
        const int tn = t + gridDim.x * WAVES;
        const int64_t rown = (int64_t)(tn < ntiles ? tn : t) * 16 + n;
        const int64_t rln = rown < a.n_rows ? rown : a.n_rows - 1;
        k0n = a.row_ptr[rln];
        k1n = a.row_ptr[rln + 1];
      }
      const int t0 = k0 >> 4, t1 = (k1 - 1) >> 4;
      const bool single = t0 == t1;
      const int nsrc = (k1 <= k0) ? 0 : (single ? 1 : t1 - t0 + 1);
#pragma unroll
      for (int mb = 0; mb < NMB; ++mb) X1[mb] = f32x4{0.f, 0.f, 0.f, 0.f};
      for (int s = 0; __any(s < nsrc); ++s) {
        if (s < nsrc) {
          const int tt = t0 + s;
          const float* src = single ? a.agg + rl * 128
                                    : a.part + ((int64_t)tt * 2 + (k0 <= (tt << 4) ? 0 : 1)) * 128;
          const f32x4* s4 = reinterpret_cast<const f32x4*>(src) + g;
#pragma unroll
          for (int mb = 0; mb < NMB; ++mb) X1[mb] = X1[mb] + s4[4 * mb];
        }
      }
    }
    const f32x4 na = reinterpret_cast<const f32x4*>(a.nattr)[rl];
    const float at[3] = {na[1], na[2], na[3]};
    f32x4 S[4], T[2], V[3][2];
#pragma unroll
    for (int m = 0; m < 4; ++m) S[m] = vec[4 * m];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      T[m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < 3; ++c) V[c][m] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    sg_operand<NC, 4, 4, true>(w0 + SGU_WS0, w0 + SGU_WT0, w0 + SGU_WV0, X0, at, S, T, V);
    sg_operand<NC, 4, 4, true>(w0 + SGU_WS0 + 2 * 512, w0 + SGU_WT0 + 256, w0 + SGU_WV0 + 256, X1, at, S, T, V);
    f32x4 H[8];
    sg_gate<NC>(S, T, V, at, H);
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      S[m] = vec[16 + 4 * m];
      T[m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < 3; ++c) V[c][m] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    sg_operand<NC, 2, 2, true>(w0 + SGU_WS1, w0 + SGU_WT1, w0 + SGU_WV1, H, at, S, T, V);
    asm volatile("" : "+v"(k0n), "+v"(k1n));
    if (valid) {  // residual, segnn.py:331 (2D: the z component stays the embedding's exact zero)
      frow[0] = X0[0] + S[0];
      frow[4] = X0[1] + S[1];
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        frow[4 * (2 + 2 * c)] = X0[2 + 2 * c] + (V[c][0] + T[0] * at[c]);
        frow[4 * (3 + 2 * c)] = X0[3 + 2 * c] + (V[c][1] + T[1] * at[c]);
      }
    }
  }
}

// ------------------------------------------------------------------------------ host side
// LDS image of one layer from the raw block weights (oracle channel order):
//   ws0 (130 x 64), wv0 (130 x 32), b0 (64); ws1 (64 x 64), wv1 (64 x 32), b1 (64).
// Folded constants: S columns * -log2(e) (the gate evaluates exp2), T / V matrices * C_sigmoid, the block-1 rows
// that consume block 0's scalars * C_silu / -log2(e) (k_sg_msg leaves them as z sigma(z)).
void lb_sg_msg_image(const float* ws0, const float* wv0, const float* b0, const float* ws1,
                     const float* wv1, const float* b1, float* out /* SGM_IMAGE*4 floats */) {
  const float sc0 = 1.0f / sqrtf(130.f), sc1 = 1.0f / sqrtf(64.f), is3 = 0.5773502691896258f;
  const float zs = SG_NL2E, cg = SG_C_SIGMOID, ks = SG_K_SILU;
  memset(out, 0, sizeof(float) * SGM_IMAGE * 4);
  std::vector<float> m;
  auto pack = [&](int K, int M, int off) { lb_pack_weight16h(m.data(), K, M, K, out + (size_t)off * 4, M); };
  // WS0: [sender s * Y0 | sender v / sqrt3 | receiver s * Y0 | receiver v / sqrt3]
  m.assign(128 * 64, 0.f);
  for (int k = 0; k < 128; ++k) {
    const float f = (((k >> 5) & 1) ? is3 * sc0 : SG_Y0 * sc0) * zs;
    for (int j = 0; j < 64; ++j) m[k * 64 + j] = ws0[k * 64 + j] * f;
  }
  pack(128, 64, SGM_WS0);
  // WT0: s rows of wv0 (sender, receiver); WV0: v rows * Y0
  m.assign(64 * 32, 0.f);
  for (int o = 0; o < 2; ++o)
    for (int k = 0; k < 32; ++k)
      for (int j = 0; j < 32; ++j) m[(o * 32 + k) * 32 + j] = wv0[(o * 64 + k) * 32 + j] * (sc0 * cg);
  pack(64, 32, SGM_WT0);
  for (int o = 0; o < 2; ++o)
    for (int k = 0; k < 32; ++k)
      for (int j = 0; j < 32; ++j) m[(o * 32 + k) * 32 + j] = wv0[(o * 64 + 32 + k) * 32 + j] * (SG_Y0 * sc0 * cg);
  pack(64, 32, SGM_WV0);
  m.assign(64 * 64, 0.f);
  for (int k = 0; k < 64; ++k) {
    const float f = ((k >= 32) ? is3 * sc1 : SG_Y0 * sc1 * ks) * zs;
    for (int j = 0; j < 64; ++j) m[k * 64 + j] = ws1[k * 64 + j] * f;
  }
  pack(64, 64, SGM_WS1);
  m.assign(32 * 32, 0.f);
  for (int k = 0; k < 32; ++k)
    for (int j = 0; j < 32; ++j) m[k * 32 + j] = wv1[k * 32 + j] * (sc1 * ks * cg);
  pack(32, 32, SGM_WT1);
  for (int k = 0; k < 32; ++k)
    for (int j = 0; j < 32; ++j) m[k * 32 + j] = wv1[(32 + k) * 32 + j] * (SG_Y0 * sc1 * cg);
  pack(32, 32, SGM_WV1);
  float* v = out + (size_t)SGM_VEC * 4;
  for (int j = 0; j < 64; ++j) {
    v[j] = b0[j] * zs;
    v[64 + j] = ws0[128 * 64 + j] * (SG_Y0 * sc0 * zs);   // |r| row
    v[128 + j] = ws0[129 * 64 + j] * (is3 * sc0 * zs);    // (r . a) row
    v[256 + j] = b1[j] * zs;
  }
  for (int j = 0; j < 32; ++j) {
    v[192 + j] = wv0[128 * 32 + j] * (sc0 * cg);            // |r| -> T
    v[224 + j] = wv0[129 * 32 + j] * (SG_Y0 * sc0 * cg);    // r_c = a_c |r| / Y1 -> T (times |r| / Y1 in the kernel)
  }
}

int lb_sg_msg_image_floats(void) { return SGM_IMAGE * 4; }

int lbk_sg_message(lb_engine* e, const float* f, const float* image, float* agg, float* part, int64_t out_bytes,
                   bool finish) {
  lb_sg_msg_args a{};
  a.ctrl = e->ctrl;
  a.senders = e->senders;
  a.receivers = e->receivers;
  a.efeat = e->efeat;
  a.f = f;
  a.image = image;
  a.agg = agg;
  a.part = part;
  if (e->BN * 512 >= ((int64_t)1 << 32)) return lb_fail(LB_ERR_UNSUPPORTED, "segnn: more than 8 M nodes per engine");
  if (out_bytes >= ((int64_t)1 << 31) || part < agg) return lb_fail(LB_ERR_UNSUPPORTED, "segnn: aggregate buffer beyond 2 GiB");
  a.part_off = (uint32_t)((const char*)part - (const char*)agg);
  a.out_bytes = (uint32_t)out_bytes;
  // no more workgroups than the frozen capacity has tiles for (B = 1: a launch is a latency chain)
  constexpr int WPS = 3;
  const int64_t tiles_cap = ((int64_t)e->e_cap * e->g.B + 15) / 16;
  int64_t gr = (tiles_cap + WPS * 4 - 1) / (WPS * 4);
  gr = (gr + 7) / 8 * 8;
  const int grid = (int)(gr < 8 ? 8 : (gr > 256 ? 256 : gr));
  if (e->g.dim == 2)
    LB_LAUNCH_TIMED(e, (k_sg_msg<2, WPS, true>), dim3(grid), dim3(WPS * 256), a);
  else
    LB_LAUNCH_TIMED(e, (k_sg_msg<3, WPS, true>), dim3(grid), dim3(WPS * 256), a);
  if (finish) {  // consumers other than k_sg_upd want complete rows in agg
    const int nb = (int)((e->BN + 7) / 8);
    hipLaunchKernelGGL(k_sg_agg_finish, dim3(nb), dim3(256), 0, e->stream, e->ctrl, e->BN, e->row_ptr,
                       part, agg);
  }
  LB_HIP(hipGetLastError());
  return LB_OK;
}

// LDS image of one layer's update from the raw block weights (oracle channel order):
//   ws0 (128 x 64), wv0 (128 x 32), b0 (64); ws1 (64 x 32), wv1 (64 x 32), b1 (32).
void lb_sg_upd_image(const float* ws0, const float* wv0, const float* b0, const float* ws1,
                     const float* wv1, const float* b1, float* out /* SGU_IMAGE*4 floats */) {
  const float sc0 = 1.0f / sqrtf(128.f), sc1 = 1.0f / sqrtf(64.f), is3 = 0.5773502691896258f;
  const float zs = SG_NL2E, cg = SG_C_SIGMOID, ks = SG_K_SILU;
  memset(out, 0, sizeof(float) * SGU_IMAGE * 4);
  std::vector<float> m;
  auto pack = [&](int K, int M, int off) { lb_pack_weight16h(m.data(), K, M, K, out + (size_t)off * 4, M); };
  m.assign(128 * 64, 0.f);
  for (int k = 0; k < 128; ++k) {
    const float f = (((k >> 5) & 1) ? is3 * sc0 : sc0) * zs;  // node attribute a0 == 1
    for (int j = 0; j < 64; ++j) m[k * 64 + j] = ws0[k * 64 + j] * f;
  }
  pack(128, 64, SGU_WS0);
  m.assign(64 * 32, 0.f);
  for (int o = 0; o < 2; ++o)
    for (int k = 0; k < 32; ++k)
      for (int j = 0; j < 32; ++j) m[(o * 32 + k) * 32 + j] = wv0[(o * 64 + k) * 32 + j] * (sc0 * cg);
  pack(64, 32, SGU_WT0);
  for (int o = 0; o < 2; ++o)
    for (int k = 0; k < 32; ++k)
      for (int j = 0; j < 32; ++j) m[(o * 32 + k) * 32 + j] = wv0[(o * 64 + 32 + k) * 32 + j] * (sc0 * cg);
  pack(64, 32, SGU_WV0);
  m.assign(64 * 32, 0.f);
  for (int k = 0; k < 64; ++k) {
    const float f = (k >= 32) ? is3 * sc1 : sc1 * ks;
    for (int j = 0; j < 32; ++j) m[k * 32 + j] = ws1[k * 32 + j] * f;
  }
  pack(64, 32, SGU_WS1);
  m.assign(32 * 32, 0.f);
  for (int k = 0; k < 32; ++k)
    for (int j = 0; j < 32; ++j) m[k * 32 + j] = wv1[k * 32 + j] * (sc1 * ks);
  pack(32, 32, SGU_WT1);
  for (int k = 0; k < 32; ++k)
    for (int j = 0; j < 32; ++j) m[k * 32 + j] = wv1[(32 + k) * 32 + j] * sc1;
  pack(32, 32, SGU_WV1);
  float* v = out + (size_t)SGU_VEC * 4;
  for (int j = 0; j < 64; ++j) v[j] = b0[j] * zs;
  for (int j = 0; j < 32; ++j) v[64 + j] = b1[j];
}

int lb_sg_upd_image_floats(void) { return SGU_IMAGE * 4; }

int lbk_sg_update(lb_engine* e, float* f, const float* agg, const float* part, const float* nattr, const float* image,
                  bool combine_partials) {
  lb_sg_upd_args a{};
  a.ctrl = e->ctrl;
  a.n_rows = e->BN;
  a.f = f;
  a.agg = agg;
  a.nattr = nattr;
  a.image = image;
  a.row_ptr = e->row_ptr;
  a.part = combine_partials ? part : nullptr;
  const int ntiles = (int)((e->BN + 15) / 16);
  const int nb = std::min(256, (ntiles + 7) / 8);
  if (e->g.dim == 2)
    LB_LAUNCH_TIMED(e, (k_sg_upd<2, 512>), dim3(nb), dim3(512), a);
  else
    LB_LAUNCH_TIMED(e, (k_sg_upd<3, 512>), dim3(nb), dim3(512), a);
  LB_HIP(hipGetLastError());
  return LB_OK;
}
