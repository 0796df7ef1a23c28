// lb_segnn_msg.hip - the SEGNN message function of one layer as ONE kernel:
//   gather f_sender, f_receiver -> O3TensorProductGate -> O3TensorProductGate -> segment_sum
// (SEGNNLayer._message + jraph aggregation, lagrangebench/models/segnn.py:280-304,306-311; e3nn
// conventions as in oracle/segnn_oracle.py).  blocks_per_step == 2 only; other depths use the
// per-block kernel of lb_segnn.hip.
//
// Why one kernel: SEGNN keeps no edge state between layers, so with the two blocks chained in
// registers and the aggregation fused into the epilogue NOTHING edge-sized is written: per edge the
// kernel reads two 512-B node rows (L2/MALL resident) + 40 B of list data, and per receiver it
// writes one 512-B row.
//
// Arithmetic: 16-edge tiles on v_mfma_f32_16x16x32_f16 in the f16x2 split scheme of lb_edge16.hip
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
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "lb_device.h"

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
#define MFMA16H(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16((a), (b), (c), 0, 0, 0)

#define SGM_THREADS 512
#define SGM_WAVES 8

static constexpr float SG_Y0 = 0.28209479177387814f;
static constexpr float SG_Y1 = 0.4886025119029199f;
static constexpr float SG_C_SILU = 1.6765620f;
static constexpr float SG_C_SIGMOID = 1.8462292f;

// LDS image of one layer, in f32x4 units (see lb_sg_msg_image below)
#define SGM_WS0 0      // K=128 x M=64  hi|lo: 4 p x 4 mbo x 2 x 64
#define SGM_WT0 2048   // K=64  x M=32: 2 x 2 x 2 x 64
#define SGM_WV0 2560
#define SGM_WS1 3072   // K=64 x M=64: 2 x 4 x 2 x 64
#define SGM_WT1 4096   // K=32 x M=32: 1 x 2 x 2 x 64
#define SGM_WV1 4352
#define SGM_VEC 4608   // fp32 vectors: b0(16) wdS(16) wrS(16) wdT(8) wrV(8) b1(16) = 80 f32x4
#define SGM_IMAGE 4688

__device__ __forceinline__ void sg_split8(const f32x4& x0, const f32x4& x1, h8& hi, h8& lo) {
  const f32x2_t a[4] = {{x0[0], x0[1]}, {x0[2], x0[3]}, {x1[0], x1[1]}, {x1[2], x1[3]}};
  h2_t hh[4], ll[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    hh[i] = __builtin_convertvector(a[i], h2_t);
    const f32x2_t back = __builtin_convertvector(hh[i], f32x2_t);
    ll[i] = __builtin_convertvector(a[i] - back, h2_t);
  }
  hi = h8{hh[0][0], hh[0][1], hh[1][0], hh[1][1], hh[2][0], hh[2][1], hh[3][0], hh[3][1]};
  lo = h8{ll[0][0], ll[0][1], ll[1][0], ll[1][1], ll[2][0], ll[2][1], ll[3][0], ll[3][1]};
}

// MFMA issue discipline (measured on gfx950 / ROCm 7.2, see DESIGN.md "MFMA accumulate-chain hazard"):
// an accumulate chain acc = mfma(a, b, acc) whose links are issued with fewer than ~4 independent
// MFMAs in between intermittently loses a link's contribution (hipcc rotates the accumulator
// registers, vDst != SrcC, and then under-spaces the dependent v_mfma_f32_16x16x32_f16).  Every
// pass below therefore walks >= 6 independent accumulators before it touches one again.

// One tensor-product block on an operand row X (8 f32x4: s0 s1 x0 x1 y0 y1 z0 z1) with edge
// attribute a[3]; ws/wt/wv: LDS matrices packed by lb_pack_weight16h (4 / 2 / 2 output blocks),
// ps: K-step of this operand's scalar group in WS (its vector group is ps + 1), pt: its K-step in
// WT and WV.
//   group 1 (B = s):              S[0..3] += Ws(ps),  T[0..1] += Wt(pt)          6 accumulators
//   group 2 (B = v.a, vx, vy, vz): S[0..3] += Ws(ps+1), V[c][0..1] += Wv(pt)     10 accumulators
__device__ __forceinline__ void sg_operand(const f32x4* __restrict__ ws, const f32x4* __restrict__ wt,
                                           const f32x4* __restrict__ wv, int ps, int pt, int lane,
                                           const f32x4 (&X)[8], const float (&a)[3], f32x4 (&S)[4],
                                           f32x4 (&T)[2], f32x4 (&V)[3][2]) {
  {
    h8 bh, bl, sh[4], sl[4], th[2], tl[2];
    sg_split8(X[0], X[1], bh, bl);
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      sh[m] = __builtin_bit_cast(h8, ws[((ps * 4 + m) * 2 + 0) * 64 + lane]);
      sl[m] = __builtin_bit_cast(h8, ws[((ps * 4 + m) * 2 + 1) * 64 + lane]);
    }
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      th[m] = __builtin_bit_cast(h8, wt[((pt * 2 + m) * 2 + 0) * 64 + lane]);
      tl[m] = __builtin_bit_cast(h8, wt[((pt * 2 + m) * 2 + 1) * 64 + lane]);
    }
#pragma unroll
    for (int m = 0; m < 4; ++m) S[m] = MFMA16H(sl[m], bh, S[m]);
#pragma unroll
    for (int m = 0; m < 2; ++m) T[m] = MFMA16H(tl[m], bh, T[m]);
#pragma unroll
    for (int m = 0; m < 4; ++m) S[m] = MFMA16H(sh[m], bl, S[m]);
#pragma unroll
    for (int m = 0; m < 2; ++m) T[m] = MFMA16H(th[m], bl, T[m]);
#pragma unroll
    for (int m = 0; m < 4; ++m) S[m] = MFMA16H(sh[m], bh, S[m]);
#pragma unroll
    for (int m = 0; m < 2; ++m) T[m] = MFMA16H(th[m], bh, T[m]);
    __builtin_amdgcn_sched_barrier(0);
  }
  {
    h8 dh, dl, vh[3], vl[3], sh[4], sl[4], wh[2], wl[2];
    const f32x4 d0 = X[2] * a[0] + X[4] * a[1] + X[6] * a[2];
    const f32x4 d1 = X[3] * a[0] + X[5] * a[1] + X[7] * a[2];
    sg_split8(d0, d1, dh, dl);
#pragma unroll
    for (int c = 0; c < 3; ++c) sg_split8(X[2 + 2 * c], X[3 + 2 * c], vh[c], vl[c]);
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      sh[m] = __builtin_bit_cast(h8, ws[(((ps + 1) * 4 + m) * 2 + 0) * 64 + lane]);
      sl[m] = __builtin_bit_cast(h8, ws[(((ps + 1) * 4 + m) * 2 + 1) * 64 + lane]);
    }
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      wh[m] = __builtin_bit_cast(h8, wv[((pt * 2 + m) * 2 + 0) * 64 + lane]);
      wl[m] = __builtin_bit_cast(h8, wv[((pt * 2 + m) * 2 + 1) * 64 + lane]);
    }
#define SGM_PASS(SA, DB, WA, VB)                                            \
  _Pragma("unroll") for (int m = 0; m < 4; ++m) S[m] = MFMA16H(SA[m], DB, S[m]); \
  _Pragma("unroll") for (int c = 0; c < 3; ++c) _Pragma("unroll") for (int m = 0; m < 2; ++m) \
      V[c][m] = MFMA16H(WA[m], VB[c], V[c][m]);
    SGM_PASS(sl, dh, wl, vh)
    SGM_PASS(sh, dl, wh, vl)
    SGM_PASS(sh, dh, wh, vh)
#undef SGM_PASS
    __builtin_amdgcn_sched_barrier(0);
  }
}

__device__ __forceinline__ f32x4 sg_silu4(const f32x4& x) {
  f32x4 y;
#pragma unroll
  for (int j = 0; j < 4; ++j) y[j] = (SG_C_SILU * x[j]) * __builtin_amdgcn_rcpf(1.f + __expf(-x[j]));
  return y;
}
__device__ __forceinline__ f32x4 sg_sigmoid4(const f32x4& x) {
  f32x4 y;
#pragma unroll
  for (int j = 0; j < 4; ++j) y[j] = SG_C_SIGMOID * __builtin_amdgcn_rcpf(1.f + __expf(-x[j]));
  return y;
}

struct lb_sg_msg_args {
  const lb_ctrl* ctrl;
  const int32_t* senders;
  const int32_t* receivers;
  const int32_t* row_ptr;
  const float* efeat;   // [E][8]
  const float* f;       // [BN][128] hidden state
  const float* image;   // SGM_IMAGE f32x4 of this layer
  float* agg;           // [BN][128]
  float* part;          // [ceil(E/16)][2][128]
  float* msg;           // ablation (MODE & 2): [E][128]
  int32_t dim;
};

// Schedules (MODE bit 0 = no register prefetch, bit 1 = ablation: write per-edge messages instead
// of the fused aggregation):  <1, 768> (default) three waves per SIMD, each wave loads its own tile and
// the other two hide the latency (see k_edge16n);  <0, 512> two waves per SIMD with a software
// pipeline inside the wave (LB_EDGE_WAVES=2).
template <int DBG, int NT>
__global__ void __launch_bounds__(NT, NT / 256) k_sg_msg(lb_sg_msg_args a) {
  __shared__ f32x4 sW[SGM_IMAGE];
  // the control block is read first and the poison flag acted on after the weight image is staged (LDS only):
  // neither the flag nor the edge count is a round trip of its own in front of the loads
  const int poisoned = a.ctrl->overflow_step;
  const int E = a.ctrl->n_edges_total;
  const int tid = threadIdx.x;
  {
    const f32x4* src = reinterpret_cast<const f32x4*>(a.image);
    for (int i = tid; i < SGM_IMAGE; i += NT) sW[i] = src[i];
  }
  if (poisoned >= 0) return;
  __syncthreads();
  const int ntiles = (E + 15) >> 4;
  const int lane = tid & 63, wave = tid >> 6;
  const int n = lane & 15, g = lane >> 4;
  const int xcd = blockIdx.x & 7, slot = (blockIdx.x >> 3) * (NT / 64) + wave;
  const int stride = (gridDim.x >> 3) * (NT / 64);
  const int t_lo = (int)(((int64_t)ntiles * xcd) >> 3), t_hi = (int)(((int64_t)ntiles * (xcd + 1)) >> 3);
  int t = t_lo + slot;
  if (t >= t_hi) return;
  const f32x4* f4 = reinterpret_cast<const f32x4*>(a.f);
  const f32x4* ef4 = reinterpret_cast<const f32x4*>(a.efeat);
  const f32x4* vec = &sW[SGM_VEC];
  auto rowc_of = [&](int tt) -> int64_t {
    const int row = tt * 16 + n;
    return row < E ? row : E - 1;
  };

  // software pipeline: rows of the next tile + indices of the tile after it are in flight while
  // the current tile computes; every load is unconditional (clamped), see lb_edge16.hip
  const int n_iter = (t_hi - 1 - t) / stride + 1;
  const int t_last = t + (n_iter - 1) * stride;
  f32x4 fs_n[8], fr_n[8], ef_n;
  int s_n, r_n, r_pref;
  auto issue = [&](int tt, int s, int r) {
    const int64_t rc = rowc_of(tt);
    const f32x4* ps = f4 + (int64_t)s * 32 + g;
    const f32x4* pr = f4 + (int64_t)r * 32 + g;
#pragma unroll
    for (int mb = 0; mb < 8; ++mb) {
      fs_n[mb] = ps[4 * mb];
      fr_n[mb] = pr[4 * mb];
    }
    ef_n = ef4[rc * 2];
  };
  {
    const int64_t rc = rowc_of(t);
    const int s0 = a.senders[rc], r0 = a.receivers[rc];
    issue(t, s0, r0);
    r_pref = r0;
    const int64_t rn = rowc_of(min(t + stride, t_last));
    s_n = a.senders[rn];
    r_n = a.receivers[rn];
  }

  for (int it = 0; it < n_iter; ++it, t += stride) {
    f32x4 fs[8], fr[8];
#pragma unroll
    for (int mb = 0; mb < 8; ++mb) {
      fs[mb] = fs_n[mb];
      fr[mb] = fr_n[mb];
    }
    f32x4 ef = ef_n;
    int r_cur = r_pref;
    if (DBG & 1) {  // no register prefetch: load this tile now
      const int64_t rc = rowc_of(t);
      const int s0 = a.senders[rc], r0 = a.receivers[rc];
      issue(t, s0, r0);
#pragma unroll
      for (int mb = 0; mb < 8; ++mb) {
        fs[mb] = fs_n[mb];
        fr[mb] = fr_n[mb];
      }
      r_cur = r0;
    } else {
    issue(min(t + stride, t_last), s_n, r_n);
    r_pref = r_n;
    {
      const int64_t rn = rowc_of(min(t + 2 * stride, t_last));
      s_n = a.senders[rn];
      r_n = a.receivers[rn];
    }
    }
    if (DBG & 1) ef = ef_n;
    // edge attribute a = Y1 r/|r| (0 for the self edge), message features |r| (rel_dist) and r
    const float rx = ef[0], ry = ef[1], rz = a.dim == 3 ? ef[2] : 0.f, dist = a.dim == 3 ? ef[3] : ef[2];
    const float nrm = sqrtf(rx * rx + ry * ry + rz * rz);
    const float inv = nrm == 0.f ? 0.f : SG_Y1 * __builtin_amdgcn_rcpf(nrm);
    const float at[3] = {rx * inv, ry * inv, rz * inv};
    const float rr3[3] = {rx, ry, rz};
    const float dotr = rx * at[0] + ry * at[1] + rz * at[2];

    // ---- block 0: [f_s | f_r | (r, |r|)] (x) a -> gate
    f32x4 S[4], T[2], V[3][2];
#pragma unroll
    for (int m = 0; m < 4; ++m) S[m] = vec[4 * m + g] + vec[16 + 4 * m + g] * dist + vec[32 + 4 * m + g] * dotr;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      T[m] = vec[48 + 4 * m + g] * dist;
#pragma unroll
      for (int c = 0; c < 3; ++c) V[c][m] = vec[56 + 4 * m + g] * rr3[c];
    }
    sg_operand(&sW[SGM_WS0], &sW[SGM_WT0], &sW[SGM_WV0], 0, 0, lane, fs, at, S, T, V);
    sg_operand(&sW[SGM_WS0], &sW[SGM_WT0], &sW[SGM_WV0], 2, 1, lane, fr, at, S, T, V);
    f32x4 H[8];
    {
      const f32x4 g0 = sg_sigmoid4(S[2]), g1 = sg_sigmoid4(S[3]);
      H[0] = sg_silu4(S[0]);
      H[1] = sg_silu4(S[1]);
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        H[2 + 2 * c] = (V[c][0] + T[0] * at[c]) * g0;
        H[3 + 2 * c] = (V[c][1] + T[1] * at[c]) * g1;
      }
    }
    // ---- block 1: h (x) a -> gate
#pragma unroll
    for (int m = 0; m < 4; ++m) S[m] = vec[64 + 4 * m + g];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      T[m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < 3; ++c) V[c][m] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    sg_operand(&sW[SGM_WS1], &sW[SGM_WT1], &sW[SGM_WV1], 0, 0, lane, H, at, S, T, V);
    f32x4 y[8];
    {
      const f32x4 g0 = sg_sigmoid4(S[2]), g1 = sg_sigmoid4(S[3]);
      y[0] = sg_silu4(S[0]);
      y[1] = sg_silu4(S[1]);
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        y[2 + 2 * c] = (V[c][0] + T[0] * at[c]) * g0;
        y[3 + 2 * c] = (V[c][1] + T[1] * at[c]) * g1;
      }
    }
    // ---- fused segment_sum over the receiver-sorted list (same scheme as k_edge16's epilogue)
    const int row = t * 16 + n;
    const bool valid = row < E;
    if (DBG & 2) {  // ablation: plain per-edge message rows, reduced by k_segment_sum
      if (valid) {
        f32x4* mr = reinterpret_cast<f32x4*>(a.msg) + (int64_t)row * 32 + g;
#pragma unroll
        for (int mb = 0; mb < 8; ++mb) mr[4 * mb] = y[mb];
      }
      continue;
    }
    const int rr = valid ? r_cur : (-1 - n);
    const int r_prev = __builtin_amdgcn_update_dpp(-2, rr, 0x111, 0xF, 0xF, false);
    const bool head = (n == 0) || (rr != r_prev);
    const unsigned Hm = (unsigned)(__ballot(head) & 0xffffull);
    const unsigned below = Hm & ((2u << n) - 1u);
    const int segstart = 31 - __clz(below);
    const bool tail = (n == 15) || ((Hm >> (n + 1)) & 1u);
    const float m1 = (n >= 1 && segstart <= n - 1) ? 1.f : 0.f, m2 = (n >= 2 && segstart <= n - 2) ? 1.f : 0.f;
    const float m4 = (n >= 4 && segstart <= n - 4) ? 1.f : 0.f, m8 = (n >= 8 && segstart <= n - 8) ? 1.f : 0.f;
#pragma unroll
    for (int mb = 0; mb < 8; ++mb)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float x = valid ? y[mb][j] : 0.f;
        x = __builtin_fmaf(__builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x111, 0xF, 0xF, true)), m1, x);
        x = __builtin_fmaf(__builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x112, 0xF, 0xF, true)), m2, x);
        x = __builtin_fmaf(__builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x114, 0xF, 0xF, true)), m4, x);
        x = __builtin_fmaf(__builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x118, 0xF, 0xF, true)), m8, x);
        y[mb][j] = x;
      }
    if (tail && valid) {
      const int k0 = a.row_ptr[rr], k1 = a.row_ptr[rr + 1];
      const bool complete = (k0 >> 4) == ((k1 - 1) >> 4);
      float* dst = complete ? a.agg + (int64_t)rr * 128
                            : a.part + ((int64_t)t * 2 + (k0 <= t * 16 ? 0 : 1)) * 128;
      f32x4* d4 = reinterpret_cast<f32x4*>(dst) + g;
#pragma unroll
      for (int mb = 0; mb < 8; ++mb) d4[4 * mb] = y[mb];
    }
  }
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
#define SGU_WS1 3072   // K=64 x M=64 (columns 32..63 zero: the last block has no gates)
#define SGU_WT1 4096
#define SGU_WV1 4352
#define SGU_VEC 4608   // b0 (16 f32x4), b1 (16, upper half zero)
#define SGU_IMAGE 4640

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

__global__ void __launch_bounds__(SGM_THREADS, 2) k_sg_upd(lb_sg_upd_args a) {
  __shared__ f32x4 sW[SGU_IMAGE];
  const int poisoned = a.ctrl->overflow_step;  // acted on after the staging loads are in flight, see k_sg_msg
  const int tid = threadIdx.x;
  {
    const f32x4* src = reinterpret_cast<const f32x4*>(a.image);
    for (int i = tid; i < SGU_IMAGE; i += SGM_THREADS) sW[i] = src[i];
  }
  if (poisoned >= 0) return;
  __syncthreads();
  const int lane = tid & 63, wave = tid >> 6;
  const int n = lane & 15, g = lane >> 4;
  const int ntiles = (int)((a.n_rows + 15) >> 4);
  const f32x4* vec = &sW[SGU_VEC];
  for (int t = blockIdx.x * SGM_WAVES + wave; t < ntiles; t += gridDim.x * SGM_WAVES) {
    const int64_t row = (int64_t)t * 16 + n;
    const bool valid = row < a.n_rows;
    const int64_t rl = valid ? row : a.n_rows - 1;
    f32x4* frow = reinterpret_cast<f32x4*>(a.f) + rl * 32 + g;
    const f32x4* arow = reinterpret_cast<const f32x4*>(a.agg) + rl * 32 + g;
    f32x4 X0[8], X1[8];
#pragma unroll
    for (int mb = 0; mb < 8; ++mb) X0[mb] = frow[4 * mb];
    if (a.part == nullptr) {
#pragma unroll
      for (int mb = 0; mb < 8; ++mb) X1[mb] = arow[4 * mb];
    } else {
      // aggregated messages straight from k_sg_msg: whole rows sit in agg, rows cut by a 16-edge
      // tile boundary are the sum of their per-tile partial slots in tile order (deterministic)
      const int E = a.ctrl->n_edges_total;
      int k0 = a.row_ptr[rl], k1 = a.row_ptr[rl + 1];
      k0 = k0 < E ? k0 : E;
      k1 = k1 < E ? k1 : E;
      const int t0 = k0 >> 4, t1 = (k1 - 1) >> 4;
      const bool single = t0 == t1;
      const int nsrc = (k1 <= k0) ? 0 : (single ? 1 : t1 - t0 + 1);
#pragma unroll
      for (int mb = 0; mb < 8; ++mb) X1[mb] = f32x4{0.f, 0.f, 0.f, 0.f};
      for (int s = 0; __any(s < nsrc); ++s) {
        if (s < nsrc) {
          const int tt = t0 + s;
          const float* src = single ? a.agg + rl * 128
                                    : a.part + ((int64_t)tt * 2 + (k0 <= (tt << 4) ? 0 : 1)) * 128;
          const f32x4* s4 = reinterpret_cast<const f32x4*>(src) + g;
#pragma unroll
          for (int mb = 0; mb < 8; ++mb) X1[mb] = X1[mb] + s4[4 * mb];
        }
      }
    }
    const f32x4 na = reinterpret_cast<const f32x4*>(a.nattr)[rl];
    const float at[3] = {na[1], na[2], na[3]};
    f32x4 S[4], T[2], V[3][2];
#pragma unroll
    for (int m = 0; m < 4; ++m) S[m] = vec[4 * m + g];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      T[m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < 3; ++c) V[c][m] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    sg_operand(&sW[SGU_WS0], &sW[SGU_WT0], &sW[SGU_WV0], 0, 0, lane, X0, at, S, T, V);
    sg_operand(&sW[SGU_WS0], &sW[SGU_WT0], &sW[SGU_WV0], 2, 1, lane, X1, at, S, T, V);
    f32x4 H[8];
    {
      const f32x4 g0 = sg_sigmoid4(S[2]), g1 = sg_sigmoid4(S[3]);
      H[0] = sg_silu4(S[0]);
      H[1] = sg_silu4(S[1]);
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        H[2 + 2 * c] = (V[c][0] + T[0] * at[c]) * g0;
        H[3 + 2 * c] = (V[c][1] + T[1] * at[c]) * g1;
      }
    }
#pragma unroll
    for (int m = 0; m < 4; ++m) S[m] = vec[16 + 4 * m + g];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      T[m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < 3; ++c) V[c][m] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    sg_operand(&sW[SGU_WS1], &sW[SGU_WT1], &sW[SGU_WV1], 0, 0, lane, H, at, S, T, V);
    if (valid) {  // residual, segnn.py:331
      frow[0] = X0[0] + S[0];
      frow[4] = X0[1] + S[1];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        frow[4 * (2 + 2 * c)] = X0[2 + 2 * c] + (V[c][0] + T[0] * at[c]);
        frow[4 * (3 + 2 * c)] = X0[3 + 2 * c] + (V[c][1] + T[1] * at[c]);
      }
    }
  }
}

// ------------------------------------------------------------------------------ host side
// LDS image of one layer from the raw block weights (oracle channel order):
//   ws0 (130 x 64), wv0 (130 x 32), b0 (64); ws1 (64 x 64), wv1 (64 x 32), b1 (64).
void lb_sg_msg_image(const float* ws0, const float* wv0, const float* b0, const float* ws1,
                     const float* wv1, const float* b1, float* out /* SGM_IMAGE*4 floats */) {
  const float sc0 = 1.0f / sqrtf(130.f), sc1 = 1.0f / sqrtf(64.f), is3 = 0.5773502691896258f;
  memset(out, 0, sizeof(float) * SGM_IMAGE * 4);
  std::vector<float> m;
  auto pack = [&](int K, int M, int off) { lb_pack_weight16h(m.data(), K, M, K, out + (size_t)off * 4, M); };
  // WS0: [sender s * Y0 | sender v / sqrt3 | receiver s * Y0 | receiver v / sqrt3]
  m.assign(128 * 64, 0.f);
  for (int k = 0; k < 128; ++k) {
    const float f = ((k >> 5) & 1) ? is3 * sc0 : SG_Y0 * sc0;
    for (int j = 0; j < 64; ++j) m[k * 64 + j] = ws0[k * 64 + j] * f;
  }
  pack(128, 64, SGM_WS0);
  // WT0: s rows of wv0 (sender, receiver); WV0: v rows * Y0
  m.assign(64 * 32, 0.f);
  for (int o = 0; o < 2; ++o)
    for (int k = 0; k < 32; ++k)
      for (int j = 0; j < 32; ++j) m[(o * 32 + k) * 32 + j] = wv0[(o * 64 + k) * 32 + j] * sc0;
  pack(64, 32, SGM_WT0);
  for (int o = 0; o < 2; ++o)
    for (int k = 0; k < 32; ++k)
      for (int j = 0; j < 32; ++j) m[(o * 32 + k) * 32 + j] = wv0[(o * 64 + 32 + k) * 32 + j] * (SG_Y0 * sc0);
  pack(64, 32, SGM_WV0);
  m.assign(64 * 64, 0.f);
  for (int k = 0; k < 64; ++k) {
    const float f = (k >= 32) ? is3 * sc1 : SG_Y0 * sc1;
    for (int j = 0; j < 64; ++j) m[k * 64 + j] = ws1[k * 64 + j] * f;
  }
  pack(64, 64, SGM_WS1);
  m.assign(32 * 32, 0.f);
  for (int k = 0; k < 32; ++k)
    for (int j = 0; j < 32; ++j) m[k * 32 + j] = wv1[k * 32 + j] * sc1;
  pack(32, 32, SGM_WT1);
  for (int k = 0; k < 32; ++k)
    for (int j = 0; j < 32; ++j) m[k * 32 + j] = wv1[(32 + k) * 32 + j] * (SG_Y0 * sc1);
  pack(32, 32, SGM_WV1);
  float* v = out + (size_t)SGM_VEC * 4;
  for (int j = 0; j < 64; ++j) {
    v[j] = b0[j];
    v[64 + j] = ws0[128 * 64 + j] * (SG_Y0 * sc0);   // |r| row
    v[128 + j] = ws0[129 * 64 + j] * (is3 * sc0);    // (r . a) row
    v[256 + j] = b1[j];
  }
  for (int j = 0; j < 32; ++j) {
    v[192 + j] = wv0[128 * 32 + j] * sc0;            // |r| -> T
    v[224 + j] = wv0[129 * 32 + j] * (SG_Y0 * sc0);  // r_c -> V_c
  }
}

int lb_sg_msg_image_floats(void) { return SGM_IMAGE * 4; }

int lbk_sg_message(lb_engine* e, const float* f, const float* image, float* agg, bool finish) {
  lb_sg_msg_args a{};
  a.ctrl = e->ctrl;
  a.senders = e->senders;
  a.receivers = e->receivers;
  a.row_ptr = e->row_ptr;
  a.efeat = e->efeat;
  a.f = f;
  a.image = image;
  a.agg = agg;
  a.part = e->part;
  a.dim = e->g.dim;
  a.msg = e->msg;
  // three waves per SIMD, no register prefetch (round 1's measured best; the two-wave software-pipelined schedule and
  // the per-edge-message ablation are template modes 0 / 2 of the kernel, no longer instantiated in the product)
  hipLaunchKernelGGL((k_sg_msg<1, 768>), dim3(256), dim3(768), 0, e->stream, a);
  if (finish) {  // consumers other than k_sg_upd want complete rows in agg
    const int nb = (int)((e->BN + 7) / 8);
    hipLaunchKernelGGL(k_sg_agg_finish, dim3(nb), dim3(256), 0, e->stream, e->ctrl, e->BN, e->row_ptr,
                       e->part, agg);
  }
  LB_HIP(hipGetLastError());
  return LB_OK;
}

// LDS image of one layer's update from the raw block weights (oracle channel order):
//   ws0 (128 x 64), wv0 (128 x 32), b0 (64); ws1 (64 x 32), wv1 (64 x 32), b1 (32).
void lb_sg_upd_image(const float* ws0, const float* wv0, const float* b0, const float* ws1,
                     const float* wv1, const float* b1, float* out /* SGU_IMAGE*4 floats */) {
  const float sc0 = 1.0f / sqrtf(128.f), sc1 = 1.0f / sqrtf(64.f), is3 = 0.5773502691896258f;
  memset(out, 0, sizeof(float) * SGU_IMAGE * 4);
  std::vector<float> m;
  auto pack = [&](int K, int M, int off) { lb_pack_weight16h(m.data(), K, M, K, out + (size_t)off * 4, M); };
  m.assign(128 * 64, 0.f);
  for (int k = 0; k < 128; ++k) {
    const float f = ((k >> 5) & 1) ? is3 * sc0 : sc0;  // node attribute a0 == 1
    for (int j = 0; j < 64; ++j) m[k * 64 + j] = ws0[k * 64 + j] * f;
  }
  pack(128, 64, SGU_WS0);
  m.assign(64 * 32, 0.f);
  for (int o = 0; o < 2; ++o)
    for (int k = 0; k < 32; ++k)
      for (int j = 0; j < 32; ++j) m[(o * 32 + k) * 32 + j] = wv0[(o * 64 + k) * 32 + j] * sc0;
  pack(64, 32, SGU_WT0);
  for (int o = 0; o < 2; ++o)
    for (int k = 0; k < 32; ++k)
      for (int j = 0; j < 32; ++j) m[(o * 32 + k) * 32 + j] = wv0[(o * 64 + 32 + k) * 32 + j] * sc0;
  pack(64, 32, SGU_WV0);
  m.assign(64 * 64, 0.f);
  for (int k = 0; k < 64; ++k) {
    const float f = (k >= 32) ? is3 * sc1 : sc1;
    for (int j = 0; j < 32; ++j) m[k * 64 + j] = ws1[k * 32 + j] * f;
  }
  pack(64, 64, SGU_WS1);
  m.assign(32 * 32, 0.f);
  for (int k = 0; k < 32; ++k)
    for (int j = 0; j < 32; ++j) m[k * 32 + j] = wv1[k * 32 + j] * sc1;
  pack(32, 32, SGU_WT1);
  for (int k = 0; k < 32; ++k)
    for (int j = 0; j < 32; ++j) m[k * 32 + j] = wv1[(32 + k) * 32 + j] * sc1;
  pack(32, 32, SGU_WV1);
  float* v = out + (size_t)SGU_VEC * 4;
  for (int j = 0; j < 64; ++j) v[j] = b0[j];
  for (int j = 0; j < 32; ++j) v[64 + j] = b1[j];
}

int lb_sg_upd_image_floats(void) { return SGU_IMAGE * 4; }

int lbk_sg_update(lb_engine* e, float* f, const float* agg, const float* nattr, const float* image,
                  bool combine_partials) {
  lb_sg_upd_args a{};
  a.ctrl = e->ctrl;
  a.n_rows = e->BN;
  a.f = f;
  a.agg = agg;
  a.nattr = nattr;
  a.image = image;
  a.row_ptr = e->row_ptr;
  a.part = combine_partials ? e->part : nullptr;
  const int ntiles = (int)((e->BN + 15) / 16);
  const int nb = std::min(256, (ntiles + SGM_WAVES - 1) / SGM_WAVES);
  hipLaunchKernelGGL(k_sg_upd, dim3(nb), dim3(SGM_THREADS), 0, e->stream, a);
  LB_HIP(hipGetLastError());
  return LB_OK;
}
