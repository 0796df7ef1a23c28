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
#include <stdio.h>

#include <algorithm>
#include <stdlib.h>

#include "lb_device.h"
#include "lb_msplit.h"

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
// Small graphs (one trajectory per GPU) run on the M-split kernels of lb_msplit.hip.  Measured (round 3, B = 1):
// the node kernel beats the LDS-ring kernels up to the 16 k-node mark where lb_node16s takes over (TGV2D-2.5k 15.4 ->
// 9.4 us, LDC3D-8k 19.3 -> 15.5 us per launch); the edge kernels win on the 2D graphs (TGV2D 13.3 -> 13.0, RPF2D) but
// repeat the per-tile index / segment arithmetic in each of their four waves, so from ~3 k tiles the wave-per-tile
// kernel is faster (LDC3D-8k: 38 vs 44 us).  LB_MSPLIT=0 / 1 forces the choice for all of them.
static int lb_msplit_env() {
  static const int env = getenv("LB_MSPLIT") ? atoi(getenv("LB_MSPLIT")) : -1;
  return env;
}
static bool lb_use_msplit_edge(const lb_engine* e) {
  const int64_t max_tiles = 3072;
  const int env = lb_msplit_env();
  if (!e->f16x2 || !e->fused_agg || env == 0) return false;
  if (env == 1) return true;
  return ((int64_t)e->e_cap * e->g.B + 15) / 16 <= max_tiles;
}
static bool lb_use_edge_w(const lb_engine* e) {
  // (k_edge16v stays for LB_GUARD=full engines and for graphs beyond 32-bit byte offsets)
  return !e->guard_full && e->aggpart_bytes < ((int64_t)1 << 31) &&
         (int64_t)(e->e_alloc + 32) * 512 < ((int64_t)1 << 32) && e->BN * 1024 < ((int64_t)1 << 32);  // 32-bit byte offsets
}
static bool lb_use_msplit_node(const lb_engine* e) {
  const int64_t max_nodes = 16384;
  const int env = lb_msplit_env();
  if (!e->f16x2 || env == 0) return false;
  if (env == 1) return true;
  return e->BN <= max_nodes;
}

extern "C" int lb_kernel_names(lb_engine* e, char* out, int32_t cap) {
  if (!e || !out || cap < 1) return lb_fail(LB_ERR_ARG, "null argument");
  const char* edge = !e->f16x2 ? "k_edge16<PROC,f32>"
                     : !e->fused_agg ? "k_edge16<PROC,f16x2> + k_segment_sum"
                     : lb_use_msplit_edge(e) ? "k_edge_ms (M-split, f16x2, fused segment_sum)"
                     : lb_use_edge_w(e) ? "k_edge16w<2 waves/SIMD, resident latents; last layer: deferred epilogue> (PROC, f16x2, fused segment_sum)"
                                        : "k_edge16v<2 waves/SIMD, resident latents, GEMM-phase priority> (PROC, f16x2, fused segment_sum)";
  const char* node = !e->f16x2 ? "k_node_mlp<f32>" : lb_use_msplit_node(e) ? "k_node_ms (M-split, f16x2)" : "k_node16s (f16x2)";
  snprintf(out, (size_t)cap, "edge=%s;node=%s", edge, node);
  return LB_OK;
}

int lbk_gns_forward(lb_engine* e, lb_gns* g) {
  LB_TRY(lb_gns_bind(e, g));
  if (g->generic) return lbk_gns_forward_generic(e, g);
  const bool ms_e = lb_use_msplit_edge(e), ms_n = lb_use_msplit_node(e);
  const bool ms_en = ms_n, ms_ee = ms_e, ms_pe = ms_e, ms_pn = ms_n;
  hipStream_t s = e->stream;
  const int64_t BN = e->BN;
  const int ntile_n = (int)((BN + LB_TILE - 1) / LB_TILE);
  const int L = g->desc.num_mp_steps;
  int rc;

  if (!(e->feat_done && e->feat_job.xnode == e->xnode)) {  // (rollout step: written by the neighbor-search launch)
    lb_tic(e, LB_T_NODEFEAT);
    rc = lbk_node_features(e, e->xnode, g->embed, g->desc.embedding_size,
                           g->desc.num_particle_types, nullptr, nullptr, nullptr, nullptr);
    lb_toc(e);
    if (rc) return rc;
  }

  // Kernel families.  f16x2 (default): M-split kernels on small graphs, else k_edge16v / k_edge_enc16v (edges) and
  // k_node16s (nodes).  Exact fp32 (LB_MATH=f32, or the range guard's fall-back): k_edge16<*, f32> and the 32-row
  // k_node_mlp.  Stand-alone aggregation (lb_set_fused_aggregation(0)): k_edge16 writes the messages, k_segment_sum
  // adds them up.
  auto node_mlp = [&](const lb_mlp_w& w, const float* xin, int kq, bool with_agg, bool resid, int next,
                      const float* ms_img, const float* w0h, const float* w1h, bool ms, bool dec = false) -> int {
    const bool proj = next < L;
    lb_node_args a{};
    a.ctrl = e->ctrl;
    a.n_rows = BN;
    a.xin = xin;
    a.agg = e->agg;
    a.nlat = e->nlat;
    a.w0p = w.w0;
    a.b0 = w.b0;
    a.w1p = w.w1;
    a.b1 = w.b1;
    a.ln_s = w.ln_s;
    a.ln_o = w.ln_o;
    a.wpp = proj ? g->proj_w[next] : nullptr;
    a.bp = proj ? g->proj_b[next] : nullptr;
    a.psr = e->psr;
    a.fused = e->fused_agg;
    a.tile_shift = 4;
    a.row_ptr = e->row_ptr;
    a.part = e->part;
    if (ms) {
      lb_nms_args m{};
      m.ctrl = e->ctrl;
      m.n_rows = BN;
      m.xin = xin;
      m.agg = e->agg;
      m.row_ptr = e->row_ptr;
      m.part = e->part;
      m.fused = e->fused_agg;
      m.nlat = e->nlat;
      m.w = ms_img;
      m.b0 = a.b0;
      m.b1 = a.b1;
      m.ln_s = a.ln_s;
      m.ln_o = a.ln_o;
      m.bp = a.bp;
      m.psr = e->psr;
      if (dec) {  // last layer: decoder (+ integrator in a rollout step) in the same launch
        m.bd0 = g->dec.b0;
        m.bd1 = g->dec.b1;
        m.dec_unscale = g->dec_unscale;
        m.acc_out = e->acc;
        m.out_dim = g->desc.out_dim;
        if (e->integ_job.on && g->desc.out_dim == e->g.dim) {
          m.integ = e->integ_job;
          e->integ_done = true;
        }
      }
      return lbk_node_ms(e, m, kq / 4, with_agg, resid, proj, dec);
    }
    if (e->f16x2)
      return lbk_node16s(e, a, w0h, w1h, proj ? g->proj_w_h2[next] : nullptr, kq / 4, with_agg ? 4 : 0, resid);
    if (with_agg)
      hipLaunchKernelGGL((k_node_mlp<16, 16, true>), dim3(ntile_n), dim3(64), 0, s, a);
    else if (kq == 4)
      hipLaunchKernelGGL((k_node_mlp<4, 0, false>), dim3(ntile_n), dim3(64), 0, s, a);
    else if (kq == 8)
      hipLaunchKernelGGL((k_node_mlp<8, 0, false>), dim3(ntile_n), dim3(64), 0, s, a);
    else if (kq == 12)
      hipLaunchKernelGGL((k_node_mlp<12, 0, false>), dim3(ntile_n), dim3(64), 0, s, a);
    else
      hipLaunchKernelGGL((k_node_mlp<16, 0, false>), dim3(ntile_n), dim3(64), 0, s, a);
    return LB_OK;
  };

  lb_tic(e, LB_T_ENC_NODE);
  rc = node_mlp(g->enc_node, e->xnode, g->kq_node, false, false, 0, g->ms_enc_node, g->enc_node_w0_h,
                g->enc_node_w1_h, ms_en);
  lb_toc(e);
  if (rc) return rc;

  bool decoded = false;  // the decoder ran inside the last layer's node launch
  lb_tic(e, LB_T_ENC_EDGE);
  if (ms_ee) {
    lb_ems_args m{};
    m.ctrl = e->ctrl;
    m.efeat = e->efeat;
    m.elat = e->elat;
    m.w = g->ms_enc_edge;
    m.b0 = g->enc_edge.b0;
    m.b1 = g->enc_edge.b1;
    m.ln_s = g->enc_edge.ln_s;
    m.ln_o = g->enc_edge.ln_o;
    rc = lbk_edge_enc_ms(e, m);
  } else {
    lb_edge16_args b{};
    b.ctrl = e->ctrl;
    b.efeat = e->efeat;
    b.elat = e->elat;
    b.w0p = e->f16x2 ? g->enc_edge_w0_16h : g->enc_edge_w0_16;
    b.b0 = g->enc_edge.b0;
    b.w1p = e->f16x2 ? g->enc_edge_w1_16h : g->enc_edge_w1_16;
    b.b1 = g->enc_edge.b1;
    b.ln_s = g->enc_edge.ln_s;
    b.ln_o = g->enc_edge.ln_o;
    rc = e->f16x2 ? lbk_edge_enc16v(e, b) : lbk_edge16(e, b, false, false);
  }
  lb_toc(e);
  if (rc) return rc;
  if (g->tap) LB_HIP(hipMemcpyAsync(g->tap, e->nlat, sizeof(float) * BN * LB_D, hipMemcpyDeviceToDevice, s));

  // ---- processor
  for (int k = 0; k < L; ++k) {
    const lb_mlp_w& pe = g->proc_edge[k];
    const bool skip = (k == L - 1) && e->fused_agg && !g->tap;  // the last layer's edge latents have no reader
    lb_tic_single(e, skip ? LB_T_EDGE_LAST : LB_T_EDGE_MLP);
    if (ms_pe) {
      lb_ems_args m{};
      m.ctrl = e->ctrl;
      m.senders = e->senders;
      m.receivers = e->receivers;
      m.elat = e->elat;
      m.psr = e->psr;
      m.w = g->ms_proc_edge[k];
      m.b1 = pe.b1;
      m.ln_s = pe.ln_s;
      m.ln_o = pe.ln_o;
      m.agg = e->agg;
      m.part = e->part;
      m.skip_elat_store = skip;
      rc = lbk_edge_ms(e, m);
    } else {
      lb_edge16_args b{};
      b.ctrl = e->ctrl;
      b.senders = e->senders;
      b.receivers = e->receivers;
      b.elat = e->elat;
      b.msg = e->msg;
      b.psr = e->psr;
      b.w0p = e->f16x2 ? g->proc_edge_w0_16h[k] : g->proc_edge_w0_16[k];
      b.w1p = e->f16x2 ? g->proc_edge_w1_16h[k] : g->proc_edge_w1_16[k];
      b.b1 = pe.b1;
      b.ln_s = pe.ln_s;
      b.ln_o = pe.ln_o;
      b.fused = e->fused_agg;
      b.row_ptr = e->row_ptr;
      b.agg = e->agg;
      b.part = e->part;
      b.aggpart_bytes = e->aggpart_bytes;
      b.skip_elat_store = skip;
      // round 5: k_edge16w (deferred epilogue: the previous tile's scan / aggregate stores ride in the MFMA slots) unless
      // the engine range-tests every k-group (LB_GUARD=full: k_edge16v's GUARD 2), or agg | part >= 2 GiB
      if (e->f16x2 && e->fused_agg)
        rc = lb_use_edge_w(e) ? lbk_edge16w(e, b) : lbk_edge16v(e, b);
      else
        rc = lbk_edge16(e, b, true, e->f16x2 != 0);
    }
    lb_toc(e);
    if (rc) return rc;
    if (!e->fused_agg) {
      lb_tic(e, LB_T_AGGREGATE);
      rc = lbk_segment_sum(e, e->msg, e->agg, LB_D);
      lb_toc(e);
      if (rc) return rc;
    }
    // M-split node kernel on the last layer: the decoder rides along (LB_SMALL_FUSED=0: separate k_decoder16 launch)
    const bool ms_dec_ok = lb_fused_launches();
    const bool with_dec = ms_pn && ms_dec_ok && k == L - 1 && e->f16x2 && g->desc.out_dim <= 4;
    lb_tic_single(e, LB_T_NODE_MLP);
    rc = node_mlp(g->proc_node[k], e->nlat, 16, true, true, k + 1, g->ms_proc_node[k], g->proc_node_w0_h[k],
                  g->proc_node_w1_h[k], ms_pn, with_dec);
    lb_toc(e);
    decoded = with_dec;
    if (rc) return rc;
    if (g->tap)
      LB_HIP(hipMemcpyAsync(g->tap + (size_t)(k + 1) * BN * LB_D, e->nlat, sizeof(float) * BN * LB_D,
                            hipMemcpyDeviceToDevice, s));
  }
  if (decoded) return LB_OK;
  lb_tic(e, LB_T_DECODER);
  rc = lbk_decoder16(e, g);
  lb_toc(e);
  if (rc) return rc;
  LB_HIP(hipGetLastError());
  return LB_OK;
}
