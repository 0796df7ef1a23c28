// lb_gns.hip - GNS encode / process / decode on gfx950 fp32 MFMA.
//
// Reference functions replaced (paths relative to the reference repo):
//   GNS._encoder / _processor / _decoder        lagrangebench/models/gns.py:65-133
//   build_mlp (hk.nets.MLP + hk.LayerNorm)      lagrangebench/models/utils.py:100-115
//   jraph.GraphNetwork gather / segment_sum     (3rd party) via gns.py:117-119
//
// Kernel design - "register-chained transposed MLP":
//   The network is evaluated TRANSPOSED: the weight matrix is the MFMA A operand (M = output
//   features) and a tile of 32 rows (edges or nodes) is the B operand (N = 32 rows), using
//   v_mfma_f32_32x32x2_f32 (exact fp32, 157 TF peak; gfx950 has no TF32).  In the C/D layout of
//   that instruction lane l holds, for row (l & 31), the features {32*mb + 8*q + 4*(l>>5) + j}.
//   Because the order of the K summation is free, the SAME registers are a valid B operand for
//   the next layer (k-pair = (f, f+4) across the two half-waves): Linear -> ReLU -> Linear ->
//   LayerNorm -> residual -> next projection run back-to-back out of registers with no LDS
//   transpose, no barrier and no intermediate HBM traffic.  LayerNorm over the 128 features of
//   a row is an in-lane sum of 64 registers plus ONE exchange with lane^32.
//   Weights are pre-packed on the host in fragment order ([kq][mb][lane][4], one 16-byte vector
//   per lane feeds 4 MFMAs): a weight fetch is a perfectly coalesced 1 KiB wave load from L2, or
//   a conflict-free linear ds_read_b128 when staged in LDS (the edge kernels keep both packed
//   128x128 matrices = 128 KiB of the CU's 160 KiB LDS resident and run persistently).
//   The first edge-MLP layer W0 [n_s | n_r | e] is split algebraically: the sender/receiver
//   parts are projected once per NODE (fused into the tail of the node kernel) and gathered as
//   the accumulator's initial value, which halves the dominant MFMA work.
//   Aggregation is an atomic-free segmented sum over the receiver-sorted CSR (deterministic).
#include <stdlib.h>

#include "lb_device.h"

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

// Host-side packing: W is (K, M) row-major (haiku Linear); out has Kpad*Mpad floats.
void lb_pack_weight(const float* w, int K, int M, int Kpad, int Mpad, float* out) {
  const int NKQ = Kpad / 8, NMB = Mpad / 32;
  for (int kq = 0; kq < NKQ; ++kq)
    for (int mb = 0; mb < NMB; ++mb)
      for (int lane = 0; lane < 64; ++lane)
        for (int j = 0; j < 4; ++j) {
          const int k = 8 * kq + 4 * (lane >> 5) + j;
          const int m = 32 * mb + (lane & 31);
          out[(((size_t)kq * NMB + mb) * 64 + lane) * 4 + j] =
              (k < K && m < M) ? w[(size_t)k * M + m] : 0.f;
        }
}

// acc[mb] += W^T(:, 8*NKQ k's) * B, weights fetched through `ld(kq, mb)`.
// The weight fragments of step kq+PF are fetched before the 4*NMB MFMAs of step kq are issued
// (software prefetch ring of PF+1 fragment sets; PF = 1 suffices for LDS-resident weights, the node
// kernels stream weights from L2 and use a deeper ring); sched_barrier keeps hipcc from hoisting
// every fetch of the fully unrolled loop to the top (which spills at the VGPR budget).
template <int NKQ, int NMB, int PF = 1, typename LD>
__device__ __forceinline__ void lb_gemm(LD ld, const f32x4 (&v)[NKQ], f32x16 (&acc)[NMB]) {
  f32x4 ring[PF + 1][NMB];
#pragma unroll
  for (int p = 0; p < PF; ++p)
    if (p < NKQ) {
#pragma unroll
      for (int mb = 0; mb < NMB; ++mb) ring[p][mb] = ld(p, mb);
    }
#pragma unroll
  for (int kq = 0; kq < NKQ; ++kq) {
    if (kq + PF < NKQ) {
#pragma unroll
      for (int mb = 0; mb < NMB; ++mb) ring[(kq + PF) % (PF + 1)][mb] = ld(kq + PF, mb);
    }
#pragma unroll
    for (int mb = 0; mb < NMB; ++mb) {
      const f32x4 a = ring[kq % (PF + 1)][mb];
      acc[mb] = MFMA(a[0], v[kq][0], acc[mb]);
      acc[mb] = MFMA(a[1], v[kq][1], acc[mb]);
      acc[mb] = MFMA(a[2], v[kq][2], acc[mb]);
      acc[mb] = MFMA(a[3], v[kq][3], acc[mb]);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// C-layout accumulators (4 blocks of 32 features) <-> B-operand vectors (16 groups of 4).
__device__ __forceinline__ void lb_acc_to_v(const f32x16 (&acc)[4], f32x4 (&v)[16], bool relu) {
#pragma unroll
  for (int mb = 0; mb < 4; ++mb)
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float x = acc[mb][4 * q + j];
        v[4 * mb + q][j] = relu ? fmaxf(x, 0.f) : x;
      }
}

// acc init from a per-feature vector p[128] (bias): lane half h reads f32x4 index 2*kq + h.
__device__ __forceinline__ void lb_acc_init(f32x16 (&acc)[4], const float* __restrict__ p, int h) {
  const f32x4* p4 = reinterpret_cast<const f32x4*>(p);
#pragma unroll
  for (int mb = 0; mb < 4; ++mb)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 t = p4[2 * (4 * mb + q) + h];
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[mb][4 * q + j] = t[j];
    }
}

// hk.LayerNorm(axis=-1, eps=1e-5) over the 128 features of each row, in C layout:
// y = (scale * rsqrt(var + eps)) * (x - mean) + offset, biased variance, two passes.
__device__ __forceinline__ void lb_layernorm(const f32x16 (&acc)[4], f32x4 (&y)[16],
                                             const float* __restrict__ ln_s,
                                             const float* __restrict__ ln_o, int h, const lb_ctrl* ctrl) {
  float s = 0.f;
#pragma unroll
  for (int mb = 0; mb < 4; ++mb)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[mb][r];
  s += __shfl_xor(s, 32);
  const float mean = s * ctrl->ln_inv_d;
  float vs = 0.f;
#pragma unroll
  for (int mb = 0; mb < 4; ++mb)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float d = acc[mb][r] - mean;
      vs += d * d;
    }
  vs += __shfl_xor(vs, 32);
  const float var = fmaxf(vs - ctrl->ln_pad * (mean * mean), 0.f) * ctrl->ln_inv_d;
  const float rs = 1.0f / sqrtf(var + 1e-5f);
  const f32x4* s4 = reinterpret_cast<const f32x4*>(ln_s);
  const f32x4* o4 = reinterpret_cast<const f32x4*>(ln_o);
#pragma unroll
  for (int mb = 0; mb < 4; ++mb)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 sc = s4[2 * (4 * mb + q) + h];
      const f32x4 of = o4[2 * (4 * mb + q) + h];
#pragma unroll
      for (int j = 0; j < 4; ++j)
        y[4 * mb + q][j] = (sc[j] * rs) * (acc[mb][4 * q + j] - mean) + of[j];
    }
}

// ============================================================================ edge kernels
struct lb_edge_args {
  const lb_ctrl* ctrl;
  const int32_t* senders;
  const int32_t* receivers;
  const float* efeat;  // ENC input [E][8]
  float* elat;         // [E][128] in/out
  float* msg;          // [E][128] out (PROC)
  const float* psr;    // [BN][256] = [n@Ws | n@Wr + b0]
  const float* w0p;    // packed: PROC 128x128 (edge rows of W0), ENC 8x128
  const float* b0;     // ENC only
  const float* w1p;    // packed 128x128
  const float* b1;
  const float* ln_s;
  const float* ln_o;
  // fused aggregation (PROC): receiver-sorted CSR + outputs
  int fused;
  const int32_t* row_ptr;
  float* agg;   // [BN][128] rows complete inside one tile
  float* part;  // [ntiles][2][128] segments cut by a tile boundary
};

#define EDGE_THREADS 512
#define EDGE_WAVES 8

// ABL: ablation bits for tools/edge_bench.hip only (0 in the product): 1 no Ps/Pr gather, 2 no e
// load, 4 no stores, 8 no LayerNorm, 16 no GEMM2, 32 no GEMM1.
template <bool PROC, int ABL = 0>
__global__ void __launch_bounds__(EDGE_THREADS, 2) k_edge_mlp(lb_edge_args a) {
  // PROC: [0,4096) = W0 edge part, [4096,8192) = W1.  ENC: [0,256) = W0 (K=8), [256,4352) = W1.
  constexpr int NW0 = PROC ? 4096 : 256;
  __shared__ f32x4 sW[NW0 + 4096];
  if (a.ctrl->overflow_step >= 0) return;
  const int tid = threadIdx.x;
  {
    const f32x4* g0 = reinterpret_cast<const f32x4*>(a.w0p);
    const f32x4* g1 = reinterpret_cast<const f32x4*>(a.w1p);
    for (int i = tid; i < NW0; i += EDGE_THREADS) sW[i] = g0[i];
    for (int i = tid; i < 4096; i += EDGE_THREADS) sW[NW0 + i] = g1[i];
  }
  __syncthreads();
  const int E = a.ctrl->n_edges_total;
  const int ntiles = (E + LB_TILE - 1) / LB_TILE;
  const int lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5;
  // XCD-aware tile walk: block b runs on XCD b % 8 (observed dispatch order); give every XCD one
  // contiguous eighth of the receiver-sorted edge list so its L2 keeps that range's gathers.
  const int xcd = blockIdx.x & 7, slot = (blockIdx.x >> 3) * EDGE_WAVES + wave;
  const int per_xcd = (gridDim.x >> 3) * EDGE_WAVES;
  const int t_lo = (int)(((int64_t)ntiles * xcd) >> 3), t_hi = (int)(((int64_t)ntiles * (xcd + 1)) >> 3);

  auto ld0 = [&](int kq, int mb) -> f32x4 { return sW[(kq * 4 + mb) * 64 + lane]; };
  auto ld1 = [&](int kq, int mb) -> f32x4 { return sW[NW0 + (kq * 4 + mb) * 64 + lane]; };

  for (int tile = t_lo + slot; tile < t_hi; tile += per_xcd) {
    const int row = tile * LB_TILE + (lane & 31);
    const bool valid = row < E;
    const int64_t rowc = valid ? row : (E - 1);
    f32x16 acc[4];
    f32x4 ve[16];
    f32x4* erow = reinterpret_cast<f32x4*>(a.elat) + rowc * 32 + h;
    if (PROC) {
      const int s = a.senders[rowc], r = a.receivers[rowc];
#pragma unroll
      for (int kq = 0; kq < 16; ++kq) ve[kq] = (ABL & 2) ? f32x4{1.f, 2.f, 3.f, (float)lane} : erow[2 * kq];
      const f32x4* ps = reinterpret_cast<const f32x4*>(a.psr) + (int64_t)s * 64 + h;
      const f32x4* pr = reinterpret_cast<const f32x4*>(a.psr) + (int64_t)r * 64 + 32 + h;
#pragma unroll
      for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 t = (ABL & 1) ? f32x4{0.1f, 0.2f, (float)s, (float)r}
                                    : ps[2 * (4 * mb + q)] + pr[2 * (4 * mb + q)];
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[mb][4 * q + j] = t[j];
        }
      if constexpr (!(ABL & 32)) lb_gemm<16, 4>(ld0, ve, acc);
    } else {
      f32x4 vin[1];
      vin[0] = reinterpret_cast<const f32x4*>(a.efeat)[rowc * 2 + h];
      lb_acc_init(acc, a.b0, h);
      lb_gemm<1, 4>(ld0, vin, acc);
    }
    f32x4 vh[16];
    lb_acc_to_v(acc, vh, true);
    f32x16 acc2[4];
    lb_acc_init(acc2, a.b1, h);
    if constexpr (!(ABL & 16)) lb_gemm<16, 4>(ld1, vh, acc2);
    if constexpr ((ABL & 16) != 0) {
#pragma unroll
      for (int mb = 0; mb < 4; ++mb) acc2[mb] = acc2[mb] + acc[mb];
    }
    f32x4 y[16];
    if constexpr (ABL & 8)
      lb_acc_to_v(acc2, y, false);
    else
      lb_layernorm(acc2, y, a.ln_s, a.ln_o, h, a.ctrl);
    if (valid && !((ABL & 4) && a.senders[0] != -12345)) {
      if (PROC) {
        if (!a.fused) {
          f32x4* mrow = reinterpret_cast<f32x4*>(a.msg) + rowc * 32 + h;
#pragma unroll
          for (int kq = 0; kq < 16; ++kq) mrow[2 * kq] = y[kq];  // e' for the stand-alone segment_sum
        }
        // residual (gns.py:120-122); e is re-read (L2-resident, fetched a few us ago by this CU)
        // instead of being kept in 64 VGPRs across both GEMMs
#pragma unroll
        for (int kq = 0; kq < 16; ++kq) erow[2 * kq] = erow[2 * kq] + y[kq];
      } else {
#pragma unroll
        for (int kq = 0; kq < 16; ++kq) erow[2 * kq] = y[kq];
      }
    }
    if (PROC && a.fused) {
      // ---- fused jraph.segment_sum(e', receivers): the tile's 32 edges are consecutive rows of
      // the receiver-sorted list, so every receiver is a contiguous lane range.  Segmented
      // Hillis-Steele scan across the 32 lanes of each half-wave with DPP row shifts (offsets
      // 1,2,4,8 inside a 16-lane row, row_bcast15 across rows); the last lane of a segment ends up
      // with the receiver's sum for its 64 features.  Rows lying entirely inside the tile are
      // written to agg[r]; the (at most two) segments cut by a tile boundary go to the tile's
      // partial slots and are combined, in tile order, by the node kernel.  No atomics.
      const int p = lane & 31;
      const int rr = valid ? a.receivers[rowc] : (-1 - p);
      const int r_prev = __shfl_up(rr, 1, 32);
      const bool head = (p == 0) || (rr != r_prev);
      const unsigned H = (unsigned)(__ballot(head) & 0xffffffffull);  // both halves: same pattern
      const unsigned below = H & (p == 31 ? 0xffffffffu : ((2u << p) - 1u));
      const int segstart = 31 - __clz(below);
      const bool tail = (p == 31) || ((H >> (p + 1)) & 1u);
      const int pr16 = p & 15;
      const bool m1 = pr16 >= 1 && segstart <= p - 1, m2 = pr16 >= 2 && segstart <= p - 2;
      const bool m4 = pr16 >= 4 && segstart <= p - 4, m8 = pr16 >= 8 && segstart <= p - 8;
      const bool mb15 = p >= 16 && segstart <= 15;
#pragma unroll
      for (int kq = 0; kq < 16; ++kq)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float x = y[kq][j];
          float t;
          t = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x111, 0xF, 0xF, true));
          x += m1 ? t : 0.f;
          t = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x112, 0xF, 0xF, true));
          x += m2 ? t : 0.f;
          t = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x114, 0xF, 0xF, true));
          x += m4 ? t : 0.f;
          t = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x118, 0xF, 0xF, true));
          x += m8 ? t : 0.f;
          t = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x142, 0xA, 0xF, false));
          x += mb15 ? t : 0.f;
          y[kq][j] = x;
        }
      if (tail && valid) {
        const int k0 = a.row_ptr[rr], k1 = a.row_ptr[rr + 1];
        const bool complete = (k0 >> 5) == ((k1 - 1) >> 5);
        float* dst = complete ? a.agg + (int64_t)rr * 128
                              : a.part + ((int64_t)tile * 2 + (k0 <= tile * LB_TILE ? 0 : 1)) * 128;
        f32x4* d4 = reinterpret_cast<f32x4*>(dst) + h;
#pragma unroll
        for (int kq = 0; kq < 16; ++kq) d4[2 * kq] = y[kq];
      }
    }
  }
}

// Gather a node's aggregated messages in the fused-aggregation scheme: one source when the
// receiver's CSR row lies inside a single 32-edge tile (agg[g]), else the per-tile partial slots
// in tile order.  Lane = node; v gets the lane's 64 features (half h).
__device__ __forceinline__ void lb_load_agg_fused(const int32_t* __restrict__ row_ptr,
                                                  const float* __restrict__ agg,
                                                  const float* __restrict__ part, int64_t g, int h,
                                                  int tile_shift, f32x4 (&v)[16]) {
  const int k0 = row_ptr[g], k1 = row_ptr[g + 1];
  const int t0 = k0 >> tile_shift, t1 = (k1 - 1) >> tile_shift;
  const bool single = t0 == t1;
  const int nsrc = (k1 <= k0) ? 0 : (single ? 1 : t1 - t0 + 1);
#pragma unroll
  for (int kq = 0; kq < 16; ++kq) v[kq] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int s = 0; __any(s < nsrc); ++s) {
    if (s < nsrc) {
      const int t = t0 + s;
      const float* src = single ? agg + g * 128
                                : part + ((int64_t)t * 2 + (k0 <= (t << tile_shift) ? 0 : 1)) * 128;
      const f32x4* s4 = reinterpret_cast<const f32x4*>(src) + h;
#pragma unroll
      for (int kq = 0; kq < 16; ++kq) v[kq] = v[kq] + s4[2 * kq];
    }
  }
}

// ============================================================================ node kernels

#define NODE_PF 3  // weight fragments in flight ahead of the MFMAs (L2 latency ~ 1-2 groups of 16 MFMAs)

template <int NKQ_A, int NKQ_B, bool RESID>
__global__ void __launch_bounds__(64) k_node_mlp(lb_node_args a) {
  if (a.ctrl->overflow_step >= 0) return;
  const int lane = threadIdx.x & 63, h = lane >> 5;
  const int64_t row = (int64_t)blockIdx.x * LB_TILE + (lane & 31);
  const bool valid = row < a.n_rows;
  const int64_t rowc = valid ? row : a.n_rows - 1;
  const f32x4* w0 = reinterpret_cast<const f32x4*>(a.w0p);
  const f32x4* w1 = reinterpret_cast<const f32x4*>(a.w1p);

  f32x16 acc[4];
  lb_acc_init(acc, a.b0, h);
  f32x4 va[NKQ_A];
  {
    const f32x4* xr = reinterpret_cast<const f32x4*>(a.xin) + rowc * (2 * NKQ_A) + h;
#pragma unroll
    for (int kq = 0; kq < NKQ_A; ++kq) va[kq] = xr[2 * kq];
    auto ld = [&](int kq, int mb) -> f32x4 { return w0[(kq * 4 + mb) * 64 + lane]; };
    lb_gemm<NKQ_A, 4, NODE_PF>(ld, va, acc);
  }
  if constexpr (NKQ_B > 0) {
    f32x4 vb[NKQ_B];
    if (a.fused) {
      lb_load_agg_fused(a.row_ptr, a.agg, a.part, rowc, h, a.tile_shift, vb);
    } else {
      const f32x4* gr = reinterpret_cast<const f32x4*>(a.agg) + rowc * (2 * NKQ_B) + h;
#pragma unroll
      for (int kq = 0; kq < NKQ_B; ++kq) vb[kq] = gr[2 * kq];
    }
    auto ld = [&](int kq, int mb) -> f32x4 { return w0[((NKQ_A + kq) * 4 + mb) * 64 + lane]; };
    lb_gemm<NKQ_B, 4, NODE_PF>(ld, vb, acc);
  }
  f32x4 vh[16];
  lb_acc_to_v(acc, vh, true);
  f32x16 acc2[4];
  lb_acc_init(acc2, a.b1, h);
  {
    auto ld = [&](int kq, int mb) -> f32x4 { return w1[(kq * 4 + mb) * 64 + lane]; };
    lb_gemm<16, 4, NODE_PF>(ld, vh, acc2);
  }
  f32x4 y[16];
  lb_layernorm(acc2, y, a.ln_s, a.ln_o, h, a.ctrl);
  if constexpr (RESID) {
    static_assert(NKQ_A == 16, "residual needs a 128-wide input");
#pragma unroll
    for (int kq = 0; kq < 16; ++kq) y[kq] = va[kq] + y[kq];
  }
  if (valid) {
    f32x4* nr = reinterpret_cast<f32x4*>(a.nlat) + rowc * 32 + h;
#pragma unroll
    for (int kq = 0; kq < 16; ++kq) nr[2 * kq] = y[kq];
  }
  if (a.wpp) {
    const f32x4* wp = reinterpret_cast<const f32x4*>(a.wpp);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      f32x16 accp[4];
      lb_acc_init(accp, a.bp + 128 * half, h);
      auto ld = [&](int kq, int mb) -> f32x4 { return wp[(kq * 8 + half * 4 + mb) * 64 + lane]; };
      lb_gemm<16, 4, NODE_PF>(ld, y, accp);
      if (valid) {
        f32x4* pr = reinterpret_cast<f32x4*>(a.psr) + rowc * 64 + half * 32 + h;
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            f32x4 t;
#pragma unroll
            for (int j = 0; j < 4; ++j) t[j] = accp[mb][4 * q + j];
            pr[2 * (4 * mb + q)] = t;
          }
      }
    }
  }
}

// ================================================================================= decoder
struct lb_dec_args {
  const lb_ctrl* ctrl;
  int64_t n_rows;
  const float* nlat;
  const float* w0p;
  const float* b0;
  const float* w1p;  // packed 128 x 32 (out_dim padded)
  const float* b1;   // [32]
  float* acc_out;    // [rows][4]
  int out_dim;
};

__global__ void __launch_bounds__(64) k_decoder(lb_dec_args a) {
  if (a.ctrl->overflow_step >= 0) return;
  const int lane = threadIdx.x & 63, h = lane >> 5;
  const int64_t row = (int64_t)blockIdx.x * LB_TILE + (lane & 31);
  const bool valid = row < a.n_rows;
  const int64_t rowc = valid ? row : a.n_rows - 1;
  const f32x4* w0 = reinterpret_cast<const f32x4*>(a.w0p);
  const f32x4* w1 = reinterpret_cast<const f32x4*>(a.w1p);
  f32x4 vn[16];
  const f32x4* nr = reinterpret_cast<const f32x4*>(a.nlat) + rowc * 32 + h;
#pragma unroll
  for (int kq = 0; kq < 16; ++kq) vn[kq] = nr[2 * kq];
  f32x16 acc[4];
  lb_acc_init(acc, a.b0, h);
  auto ld0 = [&](int kq, int mb) -> f32x4 { return w0[(kq * 4 + mb) * 64 + lane]; };
  lb_gemm<16, 4>(ld0, vn, acc);
  f32x4 vh[16];
  lb_acc_to_v(acc, vh, true);
  f32x16 acc2[1];
  {
    // rows of the C tile are output features j + 8q + 4h: outputs 0..3 live in lanes < 32, regs 0..3
    const f32x4* b4 = reinterpret_cast<const f32x4*>(a.b1);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 t = b4[2 * q + h];
#pragma unroll
      for (int j = 0; j < 4; ++j) acc2[0][4 * q + j] = t[j];
    }
  }
  auto ld1 = [&](int kq, int mb) -> f32x4 { return w1[(kq * 1 + mb) * 64 + lane]; };
  lb_gemm<16, 1>(ld1, vh, acc2);
  if (valid && h == 0) {
    f32x4 o = {acc2[0][0], acc2[0][1], acc2[0][2], acc2[0][3]};
    reinterpret_cast<f32x4*>(a.acc_out)[rowc] = o;
    bool bad = false;
    for (int d = 0; d < a.out_dim; ++d) bad |= !(fabsf(o[d]) <= 3.0e38f);
    if (bad) atomicOr(const_cast<int32_t*>(&a.ctrl->math_flags), LB_MATH_NONFINITE);
  }
}

// ============================================================================= aggregation
// jraph.segment_sum(e', receivers, N) on the receiver-sorted CSR: one half-wave per node row,
// lane c owns the 16-byte column chunk c; the row's messages are one contiguous HBM range that
// is streamed once.  Sequential order => deterministic, no atomics.
template <int D4>  // D/4 lanes per row (32 for D=128)
__global__ void __launch_bounds__(256) k_segment_sum(const lb_ctrl* __restrict__ ctrl,
                                                    const int32_t* __restrict__ row_ptr,
                                                    const float* __restrict__ msg,
                                                    float* __restrict__ out, int64_t n_rows) {
  if (ctrl->overflow_step >= 0) return;
  const int per_block = 256 / D4;
  const int64_t row = (int64_t)blockIdx.x * per_block + threadIdx.x / D4;
  const int c = threadIdx.x % D4;
  if (row >= n_rows) return;
  const int E = ctrl->n_edges_total;
  int k0 = row_ptr[row], k1 = row_ptr[row + 1];
  k0 = k0 < E ? k0 : E;
  k1 = k1 < E ? k1 : E;
  const f32x4* m4 = reinterpret_cast<const f32x4*>(msg);
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  int k = k0;
  for (; k + 4 <= k1; k += 4) {
    const f32x4 a0 = m4[(int64_t)(k + 0) * D4 + c];
    const f32x4 a1 = m4[(int64_t)(k + 1) * D4 + c];
    const f32x4 a2 = m4[(int64_t)(k + 2) * D4 + c];
    const f32x4 a3 = m4[(int64_t)(k + 3) * D4 + c];
    s = s + a0;
    s = s + a1;
    s = s + a2;
    s = s + a3;
  }
  for (; k < k1; ++k) s = s + m4[(int64_t)k * D4 + c];
  reinterpret_cast<f32x4*>(out)[row * D4 + c] = s;
}

int lbk_segment_sum(lb_engine* e, const float* msg, float* out, int D) {
  if (D != 128) return lb_fail(LB_ERR_UNSUPPORTED, "segment_sum: D=%d not built (128 only)", D);
  const int64_t rows = e->BN;
  const int per_block = 256 / 32;
  const int nb = (int)((rows + per_block - 1) / per_block);
  hipLaunchKernelGGL((k_segment_sum<32>), dim3(nb), dim3(256), 0, e->stream, e->ctrl, e->row_ptr,
                     msg, out, rows);
  LB_HIP(hipGetLastError());
  return LB_OK;
}

// ================================================================================= forward
int lbk_gns_forward(lb_engine* e, lb_gns* g) {
  if (g->generic) return lbk_gns_forward_generic(e, g);
  hipStream_t s = e->stream;
  const int64_t BN = e->BN;
  const int ntile_n = (int)((BN + LB_TILE - 1) / LB_TILE);
  const int edge_blocks = 256;
  int rc;

  lb_tic(e, LB_T_NODEFEAT);
  rc = lbk_node_features(e, e->xnode, g->embed, g->desc.embedding_size,
                         g->desc.num_particle_types, nullptr, nullptr, nullptr, nullptr);
  lb_toc(e);
  if (rc) return rc;

  const int L = g->desc.num_mp_steps;
  {
    lb_node_args a{};
    a.ctrl = e->ctrl;
    a.n_rows = BN;
    a.xin = e->xnode;
    a.nlat = e->nlat;
    a.w0p = g->enc_node.w0;
    a.b0 = g->enc_node.b0;
    a.w1p = g->enc_node.w1;
    a.b1 = g->enc_node.b1;
    a.ln_s = g->enc_node.ln_s;
    a.ln_o = g->enc_node.ln_o;
    a.wpp = L > 0 ? g->proj_w[0] : nullptr;
    a.bp = L > 0 ? g->proj_b[0] : nullptr;
    a.psr = e->psr;
    lb_tic(e, LB_T_ENC_NODE);
    // LB_NODE_KERNEL=h selects the round-1 node kernel (lb_node16h.hip)
    // and up to 16 k nodes (a launch is then a latency chain per workgroup: lb_node16h with its loader waves
    // measures faster - TGV2D-2.5k 15 vs 24 us per launch; from 24 k nodes lb_node16s wins by 13 %)
    static const bool node_s_env = !(getenv("LB_NODE_KERNEL") && getenv("LB_NODE_KERNEL")[0] == 'h');
    static const int64_t node_s_min = getenv("LB_NODE_S_MIN") ? atoll(getenv("LB_NODE_S_MIN")) : 16385;
    const bool node_s = node_s_env && BN >= node_s_min;
    if (e->f16x2 && node_s) {
      rc = lbk_node16s(e, a, g->enc_node_w0_h, g->enc_node_w1_h, L > 0 ? g->proj_w_h2[0] : nullptr,
                       g->kq_node / 4, 0, false);
      if (rc) return rc;
    } else if (e->f16x2) {
      rc = lbk_node16h(e, a, g->enc_node_w0_h, g->enc_node_w1_h, L > 0 ? g->proj_w_h[0] : nullptr,
                       g->kq_node / 4, 0, false);
      if (rc) return rc;
    } else if (g->kq_node == 4)
      hipLaunchKernelGGL((k_node_mlp<4, 0, false>), dim3(ntile_n), dim3(64), 0, s, a);
    else if (g->kq_node == 8)
      hipLaunchKernelGGL((k_node_mlp<8, 0, false>), dim3(ntile_n), dim3(64), 0, s, a);
    else if (g->kq_node == 12)
      hipLaunchKernelGGL((k_node_mlp<12, 0, false>), dim3(ntile_n), dim3(64), 0, s, a);
    else
      hipLaunchKernelGGL((k_node_mlp<16, 0, false>), dim3(ntile_n), dim3(64), 0, s, a);
    lb_toc(e);
  }
  {
    lb_edge_args a{};
    a.ctrl = e->ctrl;
    a.efeat = e->efeat;
    a.elat = e->elat;
    a.w0p = g->enc_edge.w0;
    a.b0 = g->enc_edge.b0;
    a.w1p = g->enc_edge.w1;
    a.b1 = g->enc_edge.b1;
    a.ln_s = g->enc_edge.ln_s;
    a.ln_o = g->enc_edge.ln_o;
    lb_tic(e, LB_T_ENC_EDGE);
    if (e->edge_tile == 16) {
      lb_edge16_args b{};
      b.ctrl = a.ctrl;
      b.efeat = a.efeat;
      b.elat = a.elat;
      b.w0p = e->f16x2 ? g->enc_edge_w0_16h : g->enc_edge_w0_16;
      b.b0 = a.b0;
      b.w1p = e->f16x2 ? g->enc_edge_w1_16h : g->enc_edge_w1_16;
      b.b1 = a.b1;
      b.ln_s = a.ln_s;
      b.ln_o = a.ln_o;
      // LB_ENC_KERNEL=h: the round-1 encoder kernel (k_edge16<ENC>)
      static const bool enc_v = !(getenv("LB_ENC_KERNEL") && getenv("LB_ENC_KERNEL")[0] == 'h');
      rc = (e->f16x2 && enc_v) ? lbk_edge_enc16v(e, b) : lbk_edge16(e, b, false, e->f16x2 != 0);
      if (rc) return rc;
    } else {
      hipLaunchKernelGGL((k_edge_mlp<false>), dim3(edge_blocks), dim3(EDGE_THREADS), 0, s, a);
    }
    lb_toc(e);
  }
  if (g->tap) LB_HIP(hipMemcpyAsync(g->tap, e->nlat, sizeof(float) * BN * LB_D, hipMemcpyDeviceToDevice, s));

  for (int k = 0; k < L; ++k) {
    {
      lb_edge_args a{};
      a.ctrl = e->ctrl;
      a.senders = e->senders;
      a.receivers = e->receivers;
      a.elat = e->elat;
      a.msg = e->msg;
      a.psr = e->psr;
      a.w0p = g->proc_edge[k].w0;
      a.w1p = g->proc_edge[k].w1;
      a.b1 = g->proc_edge[k].b1;
      a.ln_s = g->proc_edge[k].ln_s;
      a.ln_o = g->proc_edge[k].ln_o;
      a.fused = e->fused_agg;
      a.row_ptr = e->row_ptr;
      a.agg = e->agg;
      a.part = e->part;
      lb_tic_single(e, LB_T_EDGE_MLP);
      if (e->edge_tile == 16) {
        lb_edge16_args b{};
        b.ctrl = a.ctrl;
        b.senders = a.senders;
        b.receivers = a.receivers;
        b.elat = a.elat;
        b.msg = a.msg;
        b.psr = a.psr;
        b.w0p = e->f16x2 ? g->proc_edge_w0_16h[k] : g->proc_edge_w0_16[k];
        b.w1p = e->f16x2 ? g->proc_edge_w1_16h[k] : g->proc_edge_w1_16[k];
        b.b1 = a.b1;
        b.ln_s = a.ln_s;
        b.ln_o = a.ln_o;
        b.fused = a.fused;
        b.row_ptr = a.row_ptr;
        b.agg = a.agg;
        b.part = a.part;
        b.skip_elat_store = (k == L - 1) && e->fused_agg && !g->tap;
        // LB_EDGE_KERNEL: "n" = round-1 k_edge16n, "v0".."v3" = k_edge16v / k_edge16p variants
        static const int ev = [] {
          const char* s = getenv("LB_EDGE_KERNEL");
          if (!s || !s[0]) return 0;
          if (s[0] == 'n') return -1;
          return (s[0] == 'v' && s[1] >= '0' && s[1] <= '6') ? s[1] - '0' : 0;
        }();
        if (e->f16x2 && e->fused_agg && ev >= 0) {
          // LB_EDGE_PINGPONG=1: layer k reads one buffer and writes the other (the stand-alone message
          // buffer is free in fused mode).  The bare stream measures ~4 % faster out of place
          // (tools/stream_bench), the kernel does not (2.767 vs 2.766 ms per step): off by default
          static const bool pingpong = getenv("LB_EDGE_PINGPONG") && getenv("LB_EDGE_PINGPONG")[0] == '1';
          if (pingpong) {
            b.elat = (k & 1) ? e->msg : e->elat;
            b.elat_out = (k & 1) ? e->elat : e->msg;
          }
          rc = lbk_edge16v(e, b, ev);
        } else {
          rc = lbk_edge16(e, b, true, e->f16x2 != 0);
        }
        if (rc) return rc;
      } else {
        hipLaunchKernelGGL((k_edge_mlp<true>), dim3(edge_blocks), dim3(EDGE_THREADS), 0, s, a);
      }
      lb_toc(e);
    }
    if (!e->fused_agg) {
      lb_tic(e, LB_T_AGGREGATE);
      rc = lbk_segment_sum(e, e->msg, e->agg, LB_D);
      lb_toc(e);
      if (rc) return rc;
    }
    {
      lb_node_args a{};
      a.ctrl = e->ctrl;
      a.n_rows = BN;
      a.xin = e->nlat;
      a.agg = e->agg;
      a.nlat = e->nlat;
      a.w0p = g->proc_node[k].w0;
      a.b0 = g->proc_node[k].b0;
      a.w1p = g->proc_node[k].w1;
      a.b1 = g->proc_node[k].b1;
      a.ln_s = g->proc_node[k].ln_s;
      a.ln_o = g->proc_node[k].ln_o;
      a.wpp = (k + 1 < L) ? g->proj_w[k + 1] : nullptr;
      a.bp = (k + 1 < L) ? g->proj_b[k + 1] : nullptr;
      a.psr = e->psr;
      a.fused = e->fused_agg;
      a.tile_shift = e->edge_tile == 16 ? 4 : 5;
      a.row_ptr = e->row_ptr;
      a.part = e->part;
      lb_tic_single(e, LB_T_NODE_MLP);
      static const bool node_s_env = !(getenv("LB_NODE_KERNEL") && getenv("LB_NODE_KERNEL")[0] == 'h');
      static const int64_t node_s_min = getenv("LB_NODE_S_MIN") ? atoll(getenv("LB_NODE_S_MIN")) : 16385;
    const bool node_s = node_s_env && BN >= node_s_min;
      if (e->f16x2 && node_s) {
        rc = lbk_node16s(e, a, g->proc_node_w0_h[k], g->proc_node_w1_h[k],
                         (k + 1 < L) ? g->proj_w_h2[k + 1] : nullptr, 4, 4, true);
        if (rc) return rc;
      } else if (e->f16x2) {
        rc = lbk_node16h(e, a, g->proc_node_w0_h[k], g->proc_node_w1_h[k],
                         (k + 1 < L) ? g->proj_w_h[k + 1] : nullptr, 4, 4, true);
        if (rc) return rc;
      } else {
        hipLaunchKernelGGL((k_node_mlp<16, 16, true>), dim3(ntile_n), dim3(64), 0, s, a);
      }
      lb_toc(e);
    }
    if (g->tap)
      LB_HIP(hipMemcpyAsync(g->tap + (size_t)(k + 1) * BN * LB_D, e->nlat, sizeof(float) * BN * LB_D,
                            hipMemcpyDeviceToDevice, s));
  }
  {
    lb_dec_args a{};
    a.ctrl = e->ctrl;
    a.n_rows = BN;
    a.nlat = e->nlat;
    a.w0p = g->dec.w0;
    a.b0 = g->dec.b0;
    a.w1p = g->dec.w1;
    a.b1 = g->dec.b1;
    a.acc_out = e->acc;
    a.out_dim = g->desc.out_dim;
    lb_tic(e, LB_T_DECODER);
    // LB_DEC_KERNEL=h: round 1's one-wave-per-tile fp32 kernel
    static const bool dec16 = !(getenv("LB_DEC_KERNEL") && getenv("LB_DEC_KERNEL")[0] == 'h');
    if (dec16 && e->edge_tile == 16) {
      rc = lbk_decoder16(e, g);
      if (rc) return rc;
    } else {
      hipLaunchKernelGGL(k_decoder, dim3(ntile_n), dim3(64), 0, s, a);
    }
    lb_toc(e);
  }
  LB_HIP(hipGetLastError());
  return LB_OK;
}
