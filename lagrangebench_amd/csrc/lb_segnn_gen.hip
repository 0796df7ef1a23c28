// lb_segnn_gen.hip - SEGNN with general irreps: lmax_hidden / lmax_attributes <= 2, norm = None | "instance" | "batch"
// (round 5; SURVEY.md section 8 row a21's remainder).  Reference: lagrangebench/models/segnn.py:44-181 (O3TensorProduct /
// O3TensorProductGate), :252-362 (SEGNNLayer incl. the e3nn BatchNorm of :303,:347-351), :365-400 (weight_balanced_irreps),
// :513-610 (transform, call).  e3nn-jax conventions as restated in oracle/segnn_irreps_oracle.py (A1 - A10: parity with
// e3nn-jax itself is unpinned).  No shipped config takes this path (they all use lmax 1 / norm None: lb_segnn.hip and the
// fused kernels of lb_segnn_msg.hip / lb_segnn_node.hip); it exists so that every switch of the reference's SEGNN runs on
// the device.
//
// One kernel per O3TensorProduct (k_sgg_tp), one 16-row tile per 256- or 512-thread workgroup:
//   1. the operands' rows (gathered through senders / receivers for the message input), the attribute rows and the
//      Clebsch-Gordan tables go to LDS;
//   2. the attribute is contracted into the Clebsch-Gordan tables ONCE per row and (l1, l2, l3) combination,
//      M[row][l1 l2 l3][m1][m3] = sum_m2 C[m1][m2][m3] a[row][l2][m2]  (C = sqrt(2 l3 + 1) * real 3j symbol; at most 11
//      combinations, 115 floats per row) - the channels of a path share it, so forming a tensor-product channel costs
//      (2 l1 + 1)(2 l3 + 1) multiply-adds instead of (2 l1 + 1)(2 l2 + 1)(2 l3 + 1);
//      per output irrep l3 the channels X[m3][row][k] = sum_m1 M[..][m1][m3] x[m1] (k in e3nn's regrouped order: x chunk
//      major, attribute l2 minor) are formed in LDS from a per-channel table built at lb_segnn_create (where x[u] sits in
//      the staged row, its component stride, 2 l1 + 1, which M block);
//   3. e3nn's Linear for that irrep is (2 l3 + 1) x ceil(mul / 16) MFMA tiles (v_mfma_f32_16x16x4_f32: exact fp32 products):
//      A = X[m3] (16 rows x K) from LDS, B = W_l3 (K x mul, zero-padded, pre-multiplied by 1 / sqrt K) from L2; the four or eight waves
//      take tiles round-robin and leave the result in an LDS row buffer in e3nn layout;
//   4. epilogue: bias on the scalars, e3nn.gate (normalised silu / sigmoid) or the residual, coalesced row stores.
// Hidden rows are stored in e3nn's own layout (chunk after chunk, (mul, 2 l + 1) row-major) with the row stride padded to 4
// floats; the node input and the message features are read straight from the [scalars | x | y | z] rows lb_segnn.hip's
// preparation kernels write (a chunk is (offset, channel stride, component stride)).
// Aggregation: k_sgg_segsum (CSR by receiver, fixed order).  BatchNorm: two ordered reductions per trajectory (mean of the
// scalars, then mean squares), folded into one per-column affine map; "batch" on the messages is applied before the
// aggregation, as the reference does.  Everything is deterministic (no atomics).
#include <cmath>
#include <complex>
#include <cstring>
#include <type_traits>
#include <vector>

#include "lb_internal.h"

namespace {

constexpr int SGG_MAX_M = 128;  // sum over the (l1, l2, l3) combinations of (2 l1 + 1)(2 l3 + 1): 115 for lmax 2
// M element decode: bits 0-11 offset of C[m1][0][m3] in the table, 12-15 d3 (stride between m2), 16-19 d2, 20-23 l2^2
__host__ __device__ constexpr int sgg_mdec(int cg0, int d3, int d2, int a0) { return cg0 | (d3 << 12) | (d2 << 16) | (a0 << 20); }
constexpr float SGG_C_SILU = 1.6765620f;     // 1 / sqrt(E[silu(z)^2]), z ~ N(0, 1)  (oracle/segnn_oracle.py A5)
constexpr float SGG_C_SIGMOID = 1.8462292f;
enum { SGG_PLAIN = 0, SGG_GATE = 1, SGG_OUTVEC = 2 };

struct sgg_chunk { int32_t off, mul, l, cs, ms; };   // x[u][m] = row[off + u * cs + m * ms]
struct sgg_opnd {
  const float* x;
  const int32_t* gather;   // row index per output row, or null
  int32_t stride, lds_off;
};
struct sgg_out {           // one output irrep of the block's Linear
  int32_t l, mul, K, K4, N16, k_off, pad0, yoff;   // K4: K padded to a multiple of 32 (one chunk of 8 MFMA k-steps); k_off: int2 entries into ktab
  int64_t w_off;           // floats into the weight blob: the matrix in MFMA fragment order (sgg_pack_w), pre-scaled by 1 / sqrt K
};
struct sgg_gated { int32_t out_off, y_off, mul, d, gate0; };
struct sgg_args {
  const lb_ctrl* ctrl;
  int64_t n_rows;
  int32_t rows_from_ctrl, n_op;
  sgg_opnd op[3];
  const float* attr;
  int32_t attr_stride, n_out;
  sgg_out out[3];
  int32_t cg_floats;
  const int32_t* ktab;       // per output irrep (sgg_out::k_off) and channel k < K4: {offset of x[u][0] in the staged row,
                             //   component stride | 2 l1 + 1 << 12 | offset of M[combination] << 16}; 2 l1 + 1 = 0: padding
  int32_t mdec[SGG_MAX_M];   // element e of a row's M block: cg offset of (m1, m2 = 0, m3) | d2 * d3 << 16 | d3... see sgg_mdec
  int32_t ms;                // M floats per row
  const float* cg;
  const float* w;
  const float* bias;       // [scalars of the tensor product's output] or null
  int32_t mode, n_act, n_gated;
  sgg_gated gated[2];
  const float* resid;      // PLAIN: rows added to the result (stride dst_stride) or null
  float* dst;
  int32_t dst_stride, dst_dim;
  int32_t xin_stride, xs, x_floats, ys, kmax4, m0;   // m0: scalar outputs of the tensor product (bias count)
};

__device__ __forceinline__ float sgg_sigmoid(float x) { return 1.f / (1.f + expf(-x)); }

// SGG_NT threads share one 16-row tile: 256 (four waves) when the Linear of an irrep is a handful of MFMA tiles (lmax_hidden
// <= 1: 4 + 6 tiles; measured 3.7 vs 4.9 ms per DAM2D forward against 512), 512 at lmax_hidden 2 (5 + 6 + 10 tiles, one workgroup
// per CU by LDS anyway: 12.5 -> 9.9 ms)
template <int SGG_NT>
__global__ void __launch_bounds__(SGG_NT) k_sgg_tp(sgg_args a) {
  extern __shared__ float lds[];
  if (a.ctrl->overflow_step >= 0) return;
  const int64_t n_rows = a.rows_from_ctrl ? (int64_t)a.ctrl->n_edges_total : a.n_rows;
  const int64_t row0 = (int64_t)blockIdx.x * 16;
  if (row0 >= n_rows) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float* xin = lds;                                  // [16][xin_stride]
  float* att = xin + 16 * a.xin_stride;              // [16][16]
  float* cgt = att + 256;                            // [cg_floats (rounded to 4)]
  int* sidx = reinterpret_cast<int*>(cgt + ((a.cg_floats + 3) & ~3));   // [3][16]
  int* kmap = sidx + 48;                             // [kmax4][2]: this irrep's slice of ktab
  float* Mt = reinterpret_cast<float*>(kmap + 2 * a.kmax4);  // [16][ms (rounded to 4)]
  constexpr int TPR = SGG_NT / 16;                   // threads per row in the row-wise phases
  const int rr = tid / TPR, rc = tid % TPR;
  float* X = Mt + 16 * ((a.ms + 3) & ~3);            // [2 l3max + 1][16][xs]
  float* Y = X + a.x_floats;                         // [16][ys]

  if (tid < 16 * a.n_op) {
    const int o = tid >> 4, r = tid & 15;
    int64_t row = row0 + r;
    row = row < n_rows ? row : n_rows - 1;
    sidx[tid] = a.op[o].gather ? a.op[o].gather[row] : (int)row;
  }
  for (int i = tid; i < a.cg_floats; i += SGG_NT) cgt[i] = a.cg[i];
  for (int i = tid; i < 16 * a.ys; i += SGG_NT) Y[i] = 0.f;
  {
    int64_t row = row0 + rr;
    row = row < n_rows ? row : n_rows - 1;
    for (int c = rc; c < a.attr_stride; c += TPR) att[rr * 16 + c] = a.attr[row * a.attr_stride + c];
  }
  __syncthreads();
  {   // the attribute contracted into the Clebsch-Gordan tables, per row and combination
    const int ms = a.ms, ms4 = (a.ms + 3) & ~3;
    for (int el = rc; el < ms; el += TPR) {
      const int r = rr;
      const int dec = a.mdec[el];
      const int d3 = (dec >> 12) & 15, d2 = (dec >> 16) & 15;
      const float* c = cgt + (dec & 0xfff);
      const float* ar = att + r * 16 + (dec >> 20);
      float t = 0.f;
      for (int m2 = 0; m2 < d2; ++m2) t += c[m2 * d3] * ar[m2];
      Mt[r * ms4 + el] = t;
    }
  }
  {   // the operands' rows: every load of every operand is out before the first LDS store (round 5 staged operand after operand:
      // up to three dependent memory round trips in a workgroup whose life is a chain of them)
    constexpr int MAXL = 4;   // 16-byte pieces per thread and operand: rows of up to MAXL * TPR * 4 floats
    f32x4 buf[3][MAXL];
    bool wide = false;
#pragma unroll
    for (int o = 0; o < 3; ++o) {
      if (o < a.n_op) {
        const int s4 = a.op[o].stride >> 2;
        wide |= s4 > MAXL * TPR;
        const f32x4* srow = reinterpret_cast<const f32x4*>(a.op[o].x) + (int64_t)sidx[o * 16 + rr] * s4;
#pragma unroll
        for (int u = 0; u < MAXL; ++u) {
          const int c = rc + u * TPR;
          buf[o][u] = srow[c < s4 ? c : 0];
        }
      }
    }
#pragma unroll
    for (int o = 0; o < 3; ++o) {
      if (o < a.n_op) {
        const int s4 = a.op[o].stride >> 2;
        float* drow = xin + rr * a.xin_stride + a.op[o].lds_off;
#pragma unroll
        for (int u = 0; u < MAXL; ++u) {
          const int c = rc + u * TPR;
          if (c < s4) *reinterpret_cast<f32x4*>(drow + 4 * c) = buf[o][u];
        }
      }
    }
    if (wide) {   // (rows wider than the register staging covers: the remaining pieces, operand after operand)
      for (int o = 0; o < a.n_op; ++o) {
        const int s4 = a.op[o].stride >> 2;
        const f32x4* srow = reinterpret_cast<const f32x4*>(a.op[o].x) + (int64_t)sidx[o * 16 + rr] * s4;
        float* drow = xin + rr * a.xin_stride + a.op[o].lds_off;
        for (int c = rc + MAXL * TPR; c < s4; c += TPR) *reinterpret_cast<f32x4*>(drow + 4 * c) = srow[c];
      }
    }
  }

  for (int oi = 0; oi < a.n_out; ++oi) {
    const sgg_out& O = a.out[oi];
    const int d3 = 2 * O.l + 1;
    if (O.K == 0 || O.mul == 0) continue;
    const int ntile = O.N16 >> 4, T = d3 * ntile, nc = O.K4 >> 5, ps = O.K4 >> 2;
    // W in fragment order (sgg_pack_w): f32x4 entry ((nt * (nc + 2) + c) * 2 + q) * 64 + lane holds, for i = 0 .. 3, the
    // operand of k-step 8 c + 4 q + i - one 16-byte load per four MFMAs, and chunks nc, nc + 1 of every column tile are zeros so
    // that the look-ahead loads need no branch.  The first two chunks of this wave's first tile are requested before the
    // channels are formed (nothing of them depends on X), chunk c + 2 while chunk c multiplies.
    const f32x4* W = reinterpret_cast<const f32x4*>(a.w + O.w_off);
    f32x4 b0, b1, n0, n1;
    {
      const int nt = wave < T ? wave % ntile : 0;
      const f32x4* wp = W + (int64_t)nt * (nc + 2) * 128 + lane;
      b0 = wp[0];
      b1 = wp[64];
      n0 = wp[128];
      n1 = wp[192];
    }
    __syncthreads();   // operands staged / the previous irrep's MFMA reads of X are done
    for (int k = tid; k < 2 * O.K4; k += SGG_NT) kmap[k] = a.ktab[2 * O.k_off + k];
    __syncthreads();
    {
      const float* xrow = xin + rr * a.xin_stride;
      const float* Mrow = Mt + rr * ((a.ms + 3) & ~3);
      // 2 l3 + 1 as a compile-time constant (round 6: five predicated multiply-adds per m1 whatever the output irrep was -
      // a third of a forward's time went into this loop)
      auto form = [&](auto d3c) {
        constexpr int D3 = decltype(d3c)::value;
        for (int k = rc; k < O.K4; k += TPR) {
          const int kx = kmap[2 * k], ki = kmap[2 * k + 1];
          const int d1 = (ki >> 12) & 15, cs = ki & 0xfff;
          const float* xr = xrow + kx;
          const float* Mr = Mrow + (ki >> 16);   // [m1][m3]
          float acc[D3];
#pragma unroll
          for (int m3 = 0; m3 < D3; ++m3) acc[m3] = 0.f;
          for (int m1 = 0; m1 < d1; ++m1) {
            const float x1 = xr[m1 * cs];
#pragma unroll
            for (int m3 = 0; m3 < D3; ++m3) acc[m3] += Mr[m1 * D3 + m3] * x1;
          }
#pragma unroll
          for (int m3 = 0; m3 < D3; ++m3) X[(m3 * 16 + rr) * a.xs + (k & 3) * ps + (k >> 2)] = acc[m3];   // plane k mod 4, position k / 4
        }
      };
      if (d3 == 1) form(std::integral_constant<int, 1>{});
      else if (d3 == 3) form(std::integral_constant<int, 3>{});
      else form(std::integral_constant<int, 5>{});
    }
    __syncthreads();
    for (int t = wave; t < T; t += SGG_NT / 64) {
      const int m3 = t / ntile, nt = t - m3 * ntile;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      // lane (row, kq) multiplies k = 4 j + kq in k-step j: its operands of four consecutive k-steps are one 16-byte LDS read
      const f32x4* xa = reinterpret_cast<const f32x4*>(X + (m3 * 16 + (lane & 15)) * a.xs + (lane >> 4) * ps);
      const f32x4* wp = W + (int64_t)nt * (nc + 2) * 128 + lane;
      if (t != wave) {
        b0 = wp[0];
        b1 = wp[64];
        n0 = wp[128];
        n1 = wp[192];
      }
      for (int c = 0; c < nc; ++c) {
        const f32x4 m0 = wp[(c + 2) * 128], m1 = wp[(c + 2) * 128 + 64];   // two chunks (16 MFMAs) ahead
        const f32x4 a0 = xa[2 * c], a1 = xa[2 * c + 1];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[i], b0[i], acc, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[i], b1[i], acc, 0, 0, 0);
        b0 = n0;
        b1 = n1;
        n0 = m0;
        n1 = m1;
      }
      const int n = nt * 16 + (lane & 15);
      if (n < O.mul) {
#pragma unroll
        for (int v = 0; v < 4; ++v) Y[(4 * (lane >> 4) + v) * a.ys + O.yoff + n * d3 + m3] = acc[v];
      }
    }
  }
  __syncthreads();

  // ---- epilogue
  if (a.mode == SGG_OUTVEC) {   // output_irreps "1x1o": the accelerations, (rows, 4) with a zero in column 3
    if (tid < 64) {
      const int r = tid >> 2, c = tid & 3;
      if (row0 + r < n_rows) a.dst[(row0 + r) * 4 + c] = c < 3 ? Y[r * a.ys + c] : 0.f;
    }
    return;
  }
  const int ds = a.dst_stride;
  for (int c = rc; c < ds; c += TPR) {
    const int r = rr;
    if (row0 + r >= n_rows) continue;
    const float* y = Y + r * a.ys;
    float val = 0.f;
    if (c < a.dst_dim) {
      if (a.mode == SGG_GATE) {
        if (c < a.n_act) {
          const float z = y[c] + a.bias[c];
          val = SGG_C_SILU * (z * sgg_sigmoid(z));
        } else {
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            if (q < a.n_gated) {
              const sgg_gated& G = a.gated[q];
              const int rel = c - G.out_off;
              if (rel >= 0 && rel < G.mul * G.d) {
                const int gi = a.n_act + G.gate0 + rel / G.d;
                val = y[G.y_off + rel] * (SGG_C_SIGMOID * sgg_sigmoid(y[gi] + a.bias[gi]));
              }
            }
          }
        }
      } else {
        val = y[c];
        if (c < a.m0) val += a.bias[c];
        if (a.resid) val += a.resid[(row0 + r) * ds + c];
      }
    }
    a.dst[(row0 + r) * ds + c] = val;
  }
}

// ---------------------------------------------------------------------------------------------- attributes (lmax 2)
constexpr float SGG_Y0 = 0.28209479177387814f;   // 1 / (2 sqrt(pi))
constexpr float SGG_Y1 = 0.4886025119029199f;    // sqrt(3 / (4 pi))
__device__ __forceinline__ void sgg_y2(float x, float y, float z, float* o) {   // unit vector (or 0) -> 5 components (A9)
  const float s15 = 3.872983346207417f * SGG_Y0, s5 = 2.23606797749979f * SGG_Y0;
  o[0] = s15 * x * z;
  o[1] = s15 * x * y;
  o[2] = s5 * (y * y - 0.5f * (x * x + z * z));
  o[3] = s15 * y * z;
  o[4] = (0.5f * s15) * (z * z - x * x);
}
// edge attributes up to l = 2 from the l <= 1 rows (Y0, Y1 u) that k_sg_edge_prep wrote; stride 12
__global__ void k_sgg_attr_edge(const lb_ctrl* __restrict__ ctrl, const float* __restrict__ eattr4, float* __restrict__ out,
                                int64_t cap) {
  if (ctrl->overflow_step >= 0) return;
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= ctrl->n_edges_total || k >= cap) return;
  const f32x4 a = reinterpret_cast<const f32x4*>(eattr4)[k];
  const float inv = 1.0f / SGG_Y1;
  float y2[5];
  sgg_y2(a[1] * inv, a[2] * inv, a[3] * inv, y2);
  f32x4* o = reinterpret_cast<f32x4*>(out) + 3 * k;
  o[0] = a;
  o[1] = f32x4{y2[0], y2[1], y2[2], y2[3]};
  o[2] = f32x4{y2[4], 0.f, 0.f, 0.f};
}
// node attributes up to l = 2 (segnn.py:556-575): SH(velocity) + mean over the incoming edges of their attributes, l = 0
// entry 1.  The velocity comes from the node rows [scalars (ns4) | vx (nv4) | vy | vz]: channels 0 .. K-1 are the history.
__global__ void k_sgg_attr_node(const lb_ctrl* __restrict__ ctrl, int64_t BN, int K, int vel_avg, int ns4, int nv4,
                                const float* __restrict__ nodesv, const int32_t* __restrict__ row_ptr,
                                const float* __restrict__ eattr12, float* __restrict__ out) {
  if (ctrl->overflow_step >= 0) return;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= BN) return;
  const float* x = nodesv + i * (ns4 + 3 * nv4) + ns4;
  float vm[3];
  for (int d = 0; d < 3; ++d) {
    float s = 0.f;
    if (vel_avg) {
      for (int t = 0; t < K; ++t) s += x[d * nv4 + t];
      if (K > 1) s = s / (float)K;
    } else {
      s = x[d * nv4 + K - 1];
    }
    vm[d] = s;
  }
  const float nrm = sqrtf(vm[0] * vm[0] + vm[1] * vm[1] + vm[2] * vm[2]);
  const float inv = nrm == 0.f ? 0.f : 1.0f / nrm;
  const float ux = vm[0] * inv, uy = vm[1] * inv, uz = vm[2] * inv;
  float sh[9];
  sh[0] = SGG_Y0; sh[1] = SGG_Y1 * ux; sh[2] = SGG_Y1 * uy; sh[3] = SGG_Y1 * uz;
  sgg_y2(ux, uy, uz, sh + 4);
  const int E = ctrl->n_edges_total;
  int k0 = row_ptr[i], k1 = row_ptr[i + 1];
  k0 = k0 < E ? k0 : E;
  k1 = k1 < E ? k1 : E;
  float acc[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int k = k0; k < k1; ++k)
    for (int c = 0; c < 9; ++c) acc[c] += eattr12[(int64_t)k * 12 + c];
  const float cnt = (float)((k1 - k0) > 1 ? (k1 - k0) : 1);
  float* o = out + i * 12;
  o[0] = 1.f;
  for (int c = 1; c < 9; ++c) o[c] = sh[c] + acc[c] / cnt;
  o[9] = o[10] = o[11] = 0.f;
}

// ---------------------------------------------------------------------------------------------- aggregation
// jraph.segment_sum over the receivers (CSR rows), rows of `stride` floats, sequential per receiver: fixed order
__global__ void __launch_bounds__(256) k_sgg_segsum(const lb_ctrl* __restrict__ ctrl, const int32_t* __restrict__ row_ptr,
                                                    const float* __restrict__ msg, float* __restrict__ out, int64_t n_rows,
                                                    int s4) {
  if (ctrl->overflow_step >= 0) return;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_rows * s4) return;
  const int64_t row = i / s4;
  const int c = (int)(i - row * s4);
  const int E = ctrl->n_edges_total;
  int k0 = row_ptr[row], k1 = row_ptr[row + 1];
  k0 = k0 < E ? k0 : E;
  k1 = k1 < E ? k1 : E;
  const f32x4* m4 = reinterpret_cast<const f32x4*>(msg);
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  for (int k = k0; k < k1; ++k) s = s + m4[(int64_t)k * s4 + c];
  reinterpret_cast<f32x4*>(out)[i] = s;
}

// ---------------------------------------------------------------------------------------------- e3nn BatchNorm (A10)
struct sgg_bn_args {
  const lb_ctrl* ctrl;
  const int32_t* row_ptr;   // edges: rows of trajectory b = [row_ptr[b N], row_ptr[(b + 1) N]) clamped to E; null: nodes
  int64_t N;
  float* x;
  int32_t stride, dim, G, n_chunk;
  int32_t ch_off[3], ch_mul[3], ch_l[3], ch_w0[3], ch_b0[3];
  const float* weight;
  const float* bias;
  float eps;
  float* part;     // [B][G][stride]
  float* mean;     // [B][stride]
  float* scale;    // [B][stride]
  float* shift;    // [B][stride]
};
__device__ __forceinline__ void sgg_bn_range(const sgg_bn_args& a, int b, int64_t* r0, int64_t* r1) {
  if (a.row_ptr) {
    const int E = a.ctrl->n_edges_total;
    int k0 = a.row_ptr[(int64_t)b * a.N], k1 = a.row_ptr[(int64_t)(b + 1) * a.N];
    *r0 = k0 < E ? k0 : E;
    *r1 = k1 < E ? k1 : E;
  } else {
    *r0 = (int64_t)b * a.N;
    *r1 = (int64_t)(b + 1) * a.N;
  }
}
// pass 0: column sums; pass 1: column sums of (x - mean)^2 (mean is zero on the non-scalar columns); grid (G, B).
// A workgroup is RL row lanes x S4 four-column groups (RL = 256 / S4): lane rl takes the rows q0 + rl, q0 + rl + RL, ..
// of its group's chunk four at a time (four independent loads in flight), the RL partial sums are added in lane order
// through LDS: a fixed order, no atomics.
__global__ void __launch_bounds__(256) k_sgg_bn_part(sgg_bn_args a, int pass) {
  if (a.ctrl->overflow_step >= 0) return;
  __shared__ f32x4 red[256];
  const int b = blockIdx.y, g = blockIdx.x;
  int64_t r0, r1;
  sgg_bn_range(a, b, &r0, &r1);
  const int64_t n = r1 - r0, chunk = (n + a.G - 1) / a.G;
  const int64_t q0 = r0 + g * chunk, q1 = q0 + chunk < r1 ? q0 + chunk : r1;
  const int S4 = a.stride >> 2;
  const int RL = S4 >= 256 ? 1 : 256 / S4;
  const int rl = threadIdx.x / S4, c4 = threadIdx.x - rl * S4;
  const bool on = rl < RL && c4 < S4;
  const f32x4* x4 = reinterpret_cast<const f32x4*>(a.x);
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (on) {
    f32x4 mu = {0.f, 0.f, 0.f, 0.f};
    if (pass) mu = reinterpret_cast<const f32x4*>(a.mean)[b * S4 + c4];
    int64_t r = q0 + rl;
    for (; r + 7 * RL < q1; r += 8 * RL) {   // eight rows in flight per lane (round 6: four - a chunk was ~11 dependent round trips)
      f32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = x4[(r + u * RL) * S4 + c4] - mu;
      if (pass) {
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = v[u] * v[u];
      }
      s = s + (((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7])));
    }
    for (; r + 3 * RL < q1; r += 4 * RL) {
      f32x4 v0 = x4[r * S4 + c4] - mu, v1 = x4[(r + RL) * S4 + c4] - mu, v2 = x4[(r + 2 * RL) * S4 + c4] - mu,
            v3 = x4[(r + 3 * RL) * S4 + c4] - mu;
      if (pass) { v0 = v0 * v0; v1 = v1 * v1; v2 = v2 * v2; v3 = v3 * v3; }
      s = s + ((v0 + v1) + (v2 + v3));
    }
    for (; r < q1; r += RL) {
      f32x4 v = x4[r * S4 + c4] - mu;
      if (pass) v = v * v;
      s = s + v;
    }
  }
  red[threadIdx.x] = s;
  __syncthreads();
  if (rl == 0 && c4 < S4) {
    for (int q = 1; q < RL; ++q) s = s + red[q * S4 + c4];
    reinterpret_cast<f32x4*>(a.part)[((int64_t)b * a.G + g) * S4 + c4] = s;
  }
}
// pass 0: mean of the scalar columns; pass 1: per-column affine map  y = x * scale + shift;  grid (B)
__global__ void __launch_bounds__(256) k_sgg_bn_fin(sgg_bn_args a, int pass) {
  if (a.ctrl->overflow_step >= 0) return;
  __shared__ float sq[1024];
  const int b = blockIdx.x;
  int64_t r0, r1;
  sgg_bn_range(a, b, &r0, &r1);
  const float n = (float)((r1 - r0) > 1 ? (r1 - r0) : 1);
  for (int c = threadIdx.x; c < a.stride; c += 256) {
    float s = 0.f;
    {   // the G partials in chunk order, eight loads in flight (round 5: one dependent load per addition - this single-block
        // launch was the slowest of the five)
      const float* pp = a.part + (int64_t)b * a.G * a.stride + c;
      int g = 0;
      for (; g + 8 <= a.G; g += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = pp[(int64_t)(g + u) * a.stride];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
      }
      for (; g < a.G; ++g) s += pp[(int64_t)g * a.stride];
    }
    if (!pass) {
      bool scalar = false;
      for (int q = 0; q < a.n_chunk; ++q)
        if (a.ch_l[q] == 0 && c >= a.ch_off[q] && c < a.ch_off[q] + a.ch_mul[q]) scalar = true;
      a.mean[b * a.stride + c] = scalar ? s / n : 0.f;
    } else {
      sq[c] = s / n;
    }
  }
  if (!pass) return;
  __syncthreads();
  for (int c = threadIdx.x; c < a.stride; c += 256) {
    float sc = 0.f, sh = 0.f;
    for (int q = 0; q < a.n_chunk; ++q) {
      const int d = 2 * a.ch_l[q] + 1, rel = c - a.ch_off[q];
      if (rel >= 0 && rel < a.ch_mul[q] * d) {
        const int u = rel / d;
        float nrm = 0.f;
        for (int m = 0; m < d; ++m) nrm += sq[a.ch_off[q] + u * d + m];
        nrm = nrm / (float)d;
        sc = a.weight[a.ch_w0[q] + u] / sqrtf(nrm + a.eps);
        if (a.ch_l[q] == 0) sh = a.bias[a.ch_b0[q] + u] - a.mean[b * a.stride + c] * sc;
      }
    }
    a.scale[b * a.stride + c] = sc;
    a.shift[b * a.stride + c] = sh;
  }
}
__global__ void k_sgg_bn_apply(sgg_bn_args a, const int32_t* __restrict__ receivers, int64_t n_rows_static) {
  if (a.ctrl->overflow_step >= 0) return;
  const int64_t n_rows = a.row_ptr ? (int64_t)a.ctrl->n_edges_total : n_rows_static;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_rows * a.stride) return;
  const int64_t r = i / a.stride;
  const int c = (int)(i - r * a.stride);
  const int b = (int)((a.row_ptr ? (int64_t)receivers[r] : r) / a.N);
  a.x[i] = a.x[i] * a.scale[b * a.stride + c] + a.shift[b * a.stride + c];
}
// instance = True on an (N, dim) array: statistics over an axis of length one - scalars become the bias, every other
// channel is normalised by its own component-mean square
__global__ void k_sgg_inorm(sgg_bn_args a, int64_t n_rows) {
  if (a.ctrl->overflow_step >= 0) return;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_rows * a.stride) return;
  const int64_t r = i / a.stride;
  const int c = (int)(i - r * a.stride);
  const float* x = a.x + r * a.stride;
  float val = 0.f;
  for (int q = 0; q < a.n_chunk; ++q) {
    const int d = 2 * a.ch_l[q] + 1, rel = c - a.ch_off[q];
    if (rel >= 0 && rel < a.ch_mul[q] * d) {
      const int u = rel / d;
      if (a.ch_l[q] == 0) {
        val = a.bias[a.ch_b0[q] + u];   // (x - mean) = 0 exactly
      } else {
        float nrm = 0.f;
        for (int m = 0; m < d; ++m) {
          const float v = x[a.ch_off[q] + u * d + m];
          nrm += v * v;
        }
        nrm = nrm / (float)d;
        val = x[c] * (a.weight[a.ch_w0[q] + u] / sqrtf(nrm + a.eps));
      }
    }
  }
  __syncthreads();   // one row per workgroup (blockDim = stride): every input of the row is read before the first store
  a.x[i] = val;
}

// ---------------------------------------------------------------------------------------------- host: 3j symbols (A7)
double sgg_fact(int n) {
  double f = 1.0;
  for (int i = 2; i <= n; ++i) f *= i;
  return f;
}
double sgg_su2_cg(int j1, int m1, int j2, int m2, int j3, int m3) {   // Racah's formula, integer spins
  if (m3 != m1 + m2) return 0.0;
  const int vmin = std::max(std::max(-j1 + j2 + m3, -j1 + m1), 0);
  const int vmax = std::min(std::min(j2 + j3 + m1, j3 - j1 + j2), j3 + m3);
  const double C = (2.0 * j3 + 1.0) * sgg_fact(j3 + j1 - j2) * sgg_fact(j3 - j1 + j2) * sgg_fact(j1 + j2 - j3) * sgg_fact(j3 + m3) *
                   sgg_fact(j3 - m3) /
                   (sgg_fact(j1 + j2 + j3 + 1) * sgg_fact(j1 - m1) * sgg_fact(j1 + m1) * sgg_fact(j2 - m2) * sgg_fact(j2 + m2));
  double S = 0.0;
  for (int v = vmin; v <= vmax; ++v) {
    const double sgn = ((v + j2 + m2) & 1) ? -1.0 : 1.0;
    S += sgn * sgg_fact(j2 + j3 + m1 - v) * sgg_fact(j1 - m1 + v) /
         (sgg_fact(v) * sgg_fact(j3 - j1 + j2 - v) * sgg_fact(j3 + m3 - v) * sgg_fact(v + j1 - j2 - m3));
  }
  return std::sqrt(C) * S;
}
typedef std::complex<double> cplx;
std::vector<cplx> sgg_real_to_complex(int l) {   // (2l+1)^2 row-major, times (-i)^l
  const int d = 2 * l + 1;
  std::vector<cplx> q((size_t)d * d, cplx(0, 0));
  const double s = 1.0 / std::sqrt(2.0);
  for (int m = -l; m < 0; ++m) {
    q[(size_t)(l + m) * d + (l - m)] = cplx(s, 0);
    q[(size_t)(l + m) * d + (l + m)] = cplx(0, -s);
  }
  q[(size_t)l * d + l] = cplx(1, 0);
  for (int m = 1; m <= l; ++m) {
    const double sg = (m & 1) ? -1.0 : 1.0;
    q[(size_t)(l + m) * d + (l + m)] = cplx(sg * s, 0);
    q[(size_t)(l + m) * d + (l - m)] = cplx(0, sg * s);
  }
  cplx ph(1, 0);
  for (int i = 0; i < l; ++i) ph *= cplx(0, -1);
  for (cplx& v : q) v *= ph;
  return q;
}
// path coefficients under "component" normalisation: sqrt(2 l3 + 1) * real 3j symbol (Frobenius norm 1), [m1][m2][m3]
std::vector<float> sgg_cg(int l1, int l2, int l3) {
  const int d1 = 2 * l1 + 1, d2 = 2 * l2 + 1, d3 = 2 * l3 + 1;
  std::vector<double> C((size_t)d1 * d2 * d3, 0.0);
  for (int m1 = -l1; m1 <= l1; ++m1)
    for (int m2 = -l2; m2 <= l2; ++m2) {
      const int m3 = m1 + m2;
      if (std::abs(m3) <= l3) C[((size_t)(l1 + m1) * d2 + (l2 + m2)) * d3 + (l3 + m3)] = sgg_su2_cg(l1, m1, l2, m2, l3, m3);
    }
  const std::vector<cplx> Q1 = sgg_real_to_complex(l1), Q2 = sgg_real_to_complex(l2), Q3 = sgg_real_to_complex(l3);
  std::vector<double> R((size_t)d1 * d2 * d3, 0.0);
  double nrm = 0.0;
  for (int j = 0; j < d1; ++j)
    for (int l = 0; l < d2; ++l)
      for (int m = 0; m < d3; ++m) {
        cplx s(0, 0);
        for (int i = 0; i < d1; ++i)
          for (int k = 0; k < d2; ++k)
            for (int n = 0; n < d3; ++n) {
              const double c = C[((size_t)i * d2 + k) * d3 + n];
              if (c != 0.0) s += Q1[(size_t)i * d1 + j] * Q2[(size_t)k * d2 + l] * std::conj(Q3[(size_t)n * d3 + m]) * c;
            }
        R[((size_t)j * d2 + l) * d3 + m] = s.real();
        nrm += s.real() * s.real();
      }
  std::vector<float> out(R.size());
  const double sc = std::sqrt((double)d3) / std::sqrt(nrm);
  for (size_t i = 0; i < R.size(); ++i) out[i] = (float)(R[i] * sc);
  return out;
}
bool sgg_path_ok(int l1, int l2, int l3) { return std::abs(l1 - l2) <= l3 && l3 <= l1 + l2 && ((l1 + l2 + l3) & 1) == 0; }

struct sgg_block {            // one O3TensorProduct on the device: everything of sgg_args that does not change per call
  sgg_args a{};
  int n_op = 0;
  size_t lds_bytes = 0;
};

}  // namespace

struct lb_sgg {
  lb_segnn_desc desc;
  lb_engine* eng = nullptr;
  int La = 1, Lh = 1, n = 0, norm = 0;
  float eps = 1e-5f;
  int hdim = 0, HS = 0, n_hch = 0;
  sgg_chunk hch[3];
  int node_ns = 0, node_nv = 0, node_ns4 = 0, node_nv4 = 0, node_stride = 0;
  float* blob = nullptr;
  int32_t* ktab = nullptr;
  const float* cg_dev = nullptr;
  int cg_floats = 0;
  std::vector<sgg_block> blocks;   // call order
  std::vector<int64_t> norm_off;   // per layer: [msg weight, msg bias (batch only)], node weight, node bias
  float *xnode = nullptr, *nodesv = nullptr, *nattr4 = nullptr, *nattr = nullptr, *f = nullptr, *agg = nullptr;
  float* tn[2] = {nullptr, nullptr};
  int64_t e_alloc = 0;
  float *eattr4 = nullptr, *eattr = nullptr, *msgsv = nullptr;
  float* te[2] = {nullptr, nullptr};
  float *bn_part = nullptr, *bn_mean = nullptr, *bn_scale = nullptr, *bn_shift = nullptr;
  int bn_G = 128;
  float* tap = nullptr;
};

namespace {

template <typename T>
int sgg_alloc(T** p, size_t n) {
  *p = nullptr;
  LB_HIP(hipMalloc((void**)p, (n ? n : 1) * sizeof(T)));
  return LB_OK;
}

int sgg_ensure_edges(lb_sgg* m) {
  lb_engine* e = m->eng;
  if (m->e_alloc >= e->e_alloc && m->eattr4) return LB_OK;
  LB_HIP(hipStreamSynchronize(e->stream));
  for (float** p : {&m->eattr4, &m->eattr, &m->msgsv, &m->te[0], &m->te[1]}) {
    if (*p) (void)hipFree(*p);
    *p = nullptr;
  }
  LB_TRY(sgg_alloc(&m->eattr4, (size_t)e->e_alloc * 4));
  if (m->La >= 2) LB_TRY(sgg_alloc(&m->eattr, (size_t)e->e_alloc * 12));
  LB_TRY(sgg_alloc(&m->msgsv, (size_t)e->e_alloc * 16));
  LB_TRY(sgg_alloc(&m->te[0], (size_t)e->e_alloc * m->HS));
  LB_TRY(sgg_alloc(&m->te[1], (size_t)e->e_alloc * m->HS));
  m->e_alloc = e->e_alloc;
  return LB_OK;
}

int sgg_launch(lb_sgg* m, const sgg_block& b, int64_t rows, bool rows_from_ctrl, const float* const* xs, const int32_t* const* gathers,
               const float* attr, const float* resid, float* dst) {
  lb_engine* e = m->eng;
  sgg_args a = b.a;
  a.ctrl = e->ctrl;
  a.n_rows = rows;
  a.rows_from_ctrl = rows_from_ctrl ? 1 : 0;
  for (int o = 0; o < b.n_op; ++o) {
    a.op[o].x = xs[o];
    a.op[o].gather = gathers ? gathers[o] : nullptr;
  }
  a.attr = attr;
  a.resid = resid;
  a.dst = dst;
  if (rows <= 0) return LB_OK;
  const unsigned nb = (unsigned)((rows + 15) / 16);
  if (m->Lh >= 2) hipLaunchKernelGGL(k_sgg_tp<512>, dim3(nb), dim3(512), b.lds_bytes, e->stream, a);
  else hipLaunchKernelGGL(k_sgg_tp<256>, dim3(nb), dim3(256), b.lds_bytes, e->stream, a);
  LB_HIP(hipGetLastError());
  return LB_OK;
}

sgg_bn_args sgg_bn_base(lb_sgg* m, int64_t off_w, int64_t off_b, float* x, bool edges) {
  lb_engine* e = m->eng;
  sgg_bn_args a{};
  a.ctrl = e->ctrl;
  a.row_ptr = edges ? e->row_ptr : nullptr;
  a.N = e->g.N;
  a.x = x;
  a.stride = m->HS;
  a.dim = m->hdim;
  a.G = m->bn_G;
  a.n_chunk = m->n_hch;
  int w0 = 0, b0 = 0;
  for (int q = 0; q < m->n_hch; ++q) {
    a.ch_off[q] = m->hch[q].off; a.ch_mul[q] = m->hch[q].mul; a.ch_l[q] = m->hch[q].l;
    a.ch_w0[q] = w0; a.ch_b0[q] = b0;
    w0 += m->hch[q].mul;
    if (m->hch[q].l == 0) b0 += m->hch[q].mul;
  }
  a.weight = m->blob + off_w;
  a.bias = m->blob + off_b;
  a.eps = m->eps;
  a.part = m->bn_part; a.mean = m->bn_mean; a.scale = m->bn_scale; a.shift = m->bn_shift;
  return a;
}
int sgg_batch_norm(lb_sgg* m, int64_t off_w, int64_t off_b, float* x, bool edges) {
  lb_engine* e = m->eng;
  hipStream_t s = e->stream;
  sgg_bn_args a = sgg_bn_base(m, off_w, off_b, x, edges);
  const dim3 gp((unsigned)a.G, (unsigned)e->g.B);
  hipLaunchKernelGGL(k_sgg_bn_part, gp, dim3(256), 0, s, a, 0);
  hipLaunchKernelGGL(k_sgg_bn_fin, dim3((unsigned)e->g.B), dim3(256), 0, s, a, 0);
  hipLaunchKernelGGL(k_sgg_bn_part, gp, dim3(256), 0, s, a, 1);
  hipLaunchKernelGGL(k_sgg_bn_fin, dim3((unsigned)e->g.B), dim3(256), 0, s, a, 1);
  const int64_t rows = edges ? (int64_t)e->e_cap * e->g.B : e->BN;
  hipLaunchKernelGGL(k_sgg_bn_apply, dim3((unsigned)((rows * a.stride + 255) / 256)), dim3(256), 0, s, a, e->receivers, e->BN);
  LB_HIP(hipGetLastError());
  return LB_OK;
}

}  // namespace

void lb_sgg_destroy(lb_sgg* m) {
  if (!m) return;
  for (void* b : {(void*)m->blob, (void*)m->ktab, (void*)m->xnode, (void*)m->nodesv, (void*)m->nattr4, (void*)m->nattr, (void*)m->f, (void*)m->agg,
                  (void*)m->tn[0], (void*)m->tn[1], (void*)m->eattr4, (void*)m->eattr, (void*)m->msgsv, (void*)m->te[0],
                  (void*)m->te[1], (void*)m->bn_part, (void*)m->bn_mean, (void*)m->bn_scale, (void*)m->bn_shift})
    if (b) (void)hipFree(b);
  delete m;
}
void lb_sgg_set_tap(lb_sgg* m, float* tap) { m->tap = tap; }
int lb_sgg_row_floats(const lb_sgg* m) { return m->HS; }

// weights_host: per O3TensorProduct in call order, per output irrep l ascending: W_l (K_l x mul_l, rows in e3nn's regrouped
// order: x chunk major, attribute l minor), then b (scalar outputs) if the output has 0e; then, with norm, per layer
// [messages: weight (channels), bias (scalars) - "batch" only], nodes: weight, bias.
int lb_sgg_create(lb_engine* e, const lb_segnn_desc* d, const float* w, int64_t n_floats, lb_sgg** out) {
  if (d->lmax_hidden < 0 || d->lmax_hidden > 2 || d->lmax_attributes < 0 || d->lmax_attributes > 2)
    return lb_fail(LB_ERR_UNSUPPORTED, "segnn: lmax_hidden %d / lmax_attributes %d not built (<= 2)", d->lmax_hidden, d->lmax_attributes);
  if (d->norm < 0 || d->norm > 2) return lb_fail(LB_ERR_ARG, "segnn: norm %d (0 none, 1 instance, 2 batch)", d->norm);
  if (d->hidden < 1 || d->hidden > 128) return lb_fail(LB_ERR_UNSUPPORTED, "segnn: hidden multiplicity %d (1 .. 128)", d->hidden);
  const int B = d->blocks_per_step, L = d->num_mp_steps, Kv = e->g.isl - 1;
  lb_sgg* m = new lb_sgg();
  m->desc = *d;
  m->eng = e;
  m->La = d->lmax_attributes; m->Lh = d->lmax_hidden; m->n = d->hidden; m->norm = d->norm;
  m->eps = d->norm_eps > 0.f ? d->norm_eps : 1e-5f;
  m->node_ns = (e->g.has_vel_mag ? Kv : 0) + (d->homogeneous ? 0 : 9);
  m->node_nv = Kv + (e->g.has_bound ? 2 : 0) + (e->g.force_kind != LB_FORCE_NONE ? 1 : 0);
  m->node_ns4 = (m->node_ns + 3) & ~3;
  m->node_nv4 = (m->node_nv + 3) & ~3;
  m->node_stride = m->node_ns4 + 3 * m->node_nv4;
  // hidden irreps n x (0e + 1o + .. ) in e3nn layout
  m->n_hch = m->Lh + 1;
  int off = 0;
  for (int l = 0; l <= m->Lh; ++l) {
    m->hch[l] = sgg_chunk{off, m->n, l, 2 * l + 1, 1};
    off += m->n * (2 * l + 1);
  }
  m->hdim = off;
  m->HS = (off + 3) & ~3;
  if (m->HS > 1024) {
    delete m;
    return lb_fail(LB_ERR_UNSUPPORTED, "segnn: hidden rows of %d floats (<= 1024)", off);
  }
  // operand templates
  std::vector<sgg_chunk> hid(m->hch, m->hch + m->n_hch);
  std::vector<sgg_chunk> nodec;   // models/utils.py:75-97: Kx1o [+2x1o] [+1x1o] [+Kx0e] [+9x0e] on the [s | x | y | z] rows
  {
    int v = 0, s = 0;
    auto vec = [&](int mul) { nodec.push_back(sgg_chunk{m->node_ns4 + v, mul, 1, 1, m->node_nv4}); v += mul; };
    auto sca = [&](int mul) { nodec.push_back(sgg_chunk{s, mul, 0, 1, 1}); s += mul; };
    vec(Kv);
    if (e->g.has_bound) vec(2);
    if (e->g.force_kind != LB_FORCE_NONE) vec(1);
    if (e->g.has_vel_mag) sca(Kv);
    if (!d->homogeneous) sca(9);
  }
  const std::vector<sgg_chunk> msgc = {sgg_chunk{4, 1, 1, 1, 4}, sgg_chunk{0, 1, 0, 1, 1}};   // "1x1o+1x0e" on k_sg_edge_prep's rows
  // Clebsch-Gordan tables
  std::vector<float> host;
  int cg_off[3][3][3];
  for (int l1 = 0; l1 <= 2; ++l1)
    for (int l2 = 0; l2 <= 2; ++l2)
      for (int l3 = 0; l3 <= 2; ++l3) {
        cg_off[l1][l2][l3] = -1;
        if (!sgg_path_ok(l1, l2, l3)) continue;
        cg_off[l1][l2][l3] = (int)host.size();
        const std::vector<float> c = sgg_cg(l1, l2, l3);
        host.insert(host.end(), c.begin(), c.end());
      }
  m->cg_floats = (int)host.size();
  host.resize((host.size() + 63) & ~(size_t)63, 0.f);
  // the rows' M blocks: one (2 l1 + 1) x (2 l3 + 1) matrix per combination that the attributes' lmax admits
  int m_off[3][3][3], ms_total = 0;
  std::vector<int32_t> mdec;
  for (int l1 = 0; l1 <= 2; ++l1)
    for (int l2 = 0; l2 <= m->La; ++l2)
      for (int l3 = 0; l3 <= 2; ++l3) {
        m_off[l1][l2][l3] = -1;
        if (cg_off[l1][l2][l3] < 0) continue;
        m_off[l1][l2][l3] = ms_total;
        const int d1 = 2 * l1 + 1, d2 = 2 * l2 + 1, d3 = 2 * l3 + 1;
        for (int m1 = 0; m1 < d1; ++m1)
          for (int m3 = 0; m3 < d3; ++m3) mdec.push_back(sgg_mdec(cg_off[l1][l2][l3] + m1 * d2 * d3 + m3, d3, d2, l2 * l2));
        ms_total += d1 * d3;
      }
  if (ms_total > SGG_MAX_M || m->cg_floats > 0xfff) {
    delete m;
    return lb_fail(LB_ERR_STATE, "segnn: Clebsch-Gordan tables larger than planned (%d, %d)", ms_total, m->cg_floats);
  }

  const float* p = w;
  const float* pend = w + n_floats;
  bool short_blob = false, bad = false;
  std::vector<int32_t> ktab_host;   // every block's channel tables, uploaded behind the float blob
  struct Op { std::vector<sgg_chunk> ch; int stride; };
  auto add_block = [&](const std::vector<Op>& ops, const std::vector<std::pair<int, int>>& outs /* (mul, l) ascending l */, int mode) {
    sgg_block b;
    sgg_args& a = b.a;
    b.n_op = a.n_op = (int)ops.size();
    int lo = 0;
    for (int o = 0; o < b.n_op; ++o) {
      a.op[o].stride = ops[o].stride;
      a.op[o].lds_off = lo;
      lo += ops[o].stride;
    }
    a.xin_stride = lo + 4;
    a.attr_stride = m->La >= 2 ? 12 : 4;
    a.n_out = (int)outs.size();
    int yoff = 0, kmax4 = 32;
    for (int oi = 0; oi < a.n_out; ++oi) {
      sgg_out& O = a.out[oi];
      O.mul = outs[oi].first;
      O.l = outs[oi].second;
      O.yoff = yoff;
      yoff += O.mul * (2 * O.l + 1);
      // the channels of this irrep in e3nn's regrouped order (x chunk major, attribute l2 minor): one ktab entry each
      int K = 0;
      std::vector<int32_t> kt;
      for (int o = 0; o < b.n_op; ++o)
        for (const sgg_chunk& c : ops[o].ch)
          for (int l2 = 0; l2 <= m->La; ++l2) {
            if (!sgg_path_ok(c.l, l2, O.l)) continue;
            for (int u = 0; u < c.mul; ++u) {
              kt.push_back(a.op[o].lds_off + c.off + u * c.cs);
              kt.push_back(c.ms | ((2 * c.l + 1) << 12) | (m_off[c.l][l2][O.l] << 16));
            }
            K += c.mul;
          }
      O.K = K;
      O.K4 = (K + 31) & ~31;
      O.N16 = (O.mul + 15) & ~15;
      kmax4 = std::max(kmax4, O.K4);
      kt.resize((size_t)2 * O.K4, 0);                      // padding channels: 2 l1 + 1 = 0 -> zeros
      O.k_off = (int32_t)(ktab_host.size() / 2);
      ktab_host.insert(ktab_host.end(), kt.begin(), kt.end());
      if (K == 0 || O.mul == 0) { O.w_off = 0; continue; }
      if (p + (size_t)K * O.mul > pend) { short_blob = true; return; }
      // MFMA fragment order: [column tile nt][chunk c < nc + 2][q < 2][lane][i < 4] = W[32 c + 16 q + 4 i + (lane >> 4)][16 nt + (lane & 15)]
      // (k-step j = 8 c + 4 q + i multiplies k = 4 j + kq); chunks nc, nc + 1 are zeros (look-ahead loads of the kernel)
      const int nc = O.K4 / 32, ntl = O.N16 / 16;
      const size_t woff = (host.size() + 63) & ~(size_t)63;
      host.resize(woff + (size_t)ntl * (nc + 2) * 512, 0.f);
      const float sc = 1.0f / sqrtf((float)K);   // e3nn Linear, "element" normalisation (A4)
      for (int nt = 0; nt < ntl; ++nt)
        for (int c = 0; c < nc; ++c)
          for (int q = 0; q < 2; ++q)
            for (int ln = 0; ln < 64; ++ln)
              for (int i = 0; i < 4; ++i) {
                const int k = 32 * c + 16 * q + 4 * i + (ln >> 4), n = 16 * nt + (ln & 15);
                if (k < K && n < O.mul)
                  host[woff + ((((size_t)nt * (nc + 2) + c) * 2 + q) * 64 + ln) * 4 + i] = p[(size_t)k * O.mul + n] * sc;
              }
      O.w_off = (int64_t)woff;
      p += (size_t)K * O.mul;
    }
    a.m0 = (a.n_out > 0 && a.out[0].l == 0) ? a.out[0].mul : 0;
    // bias (stored as an offset in `bias` until the blob is uploaded)
    size_t boff = (host.size() + 63) & ~(size_t)63;
    host.resize(boff + (size_t)std::max(a.m0, 1), 0.f);
    if (a.m0 > 0) {
      if (p + a.m0 > pend) { short_blob = true; return; }
      memcpy(host.data() + boff, p, sizeof(float) * a.m0);
      p += a.m0;
    }
    a.bias = reinterpret_cast<const float*>(boff);
    a.mode = mode;
    a.ys = yoff + 1;
    a.kmax4 = kmax4;
    a.xs = ((kmax4 + 63) / 64) * 64 + 4;   // row stride of X: = 4 (mod 64) - the 16 rows' 16-byte reads of an MFMA operand hit 64 distinct banks
    int d3max = 1;
    for (int oi = 0; oi < a.n_out; ++oi)
      if (a.out[oi].K > 0 && a.out[oi].mul > 0) d3max = std::max(d3max, 2 * a.out[oi].l + 1);
    a.x_floats = d3max * 16 * a.xs;
    a.dst_stride = m->HS;
    a.dst_dim = m->hdim;
    if (mode == SGG_GATE) {
      a.n_act = m->n;       // hidden scalars
      int gate0 = 0, yo = a.out[0].mul, ng = 0;
      for (int l = 1; l <= m->Lh; ++l) {
        a.gated[ng++] = sgg_gated{m->hch[l].off, yo, m->n, 2 * l + 1, gate0};
        yo += m->n * (2 * l + 1);
        gate0 += m->n;
      }
      a.n_gated = ng;
    }
    a.cg_floats = m->cg_floats;
    a.ms = ms_total;
    for (int i = 0; i < ms_total; ++i) a.mdec[i] = mdec[(size_t)i];
    b.lds_bytes = sizeof(float) * ((size_t)16 * a.xin_stride + 256 + ((a.cg_floats + 3) & ~3) + 48 +
                                   2 * a.kmax4 + (size_t)16 * ((ms_total + 3) & ~3) + a.x_floats + (size_t)16 * a.ys);
    m->blocks.push_back(b);
  };
  std::vector<std::pair<int, int>> hid_out, gate_out;
  for (int l = 0; l <= m->Lh; ++l) hid_out.push_back({m->n, l});
  gate_out.push_back({m->n + m->n * m->Lh, 0});
  for (int l = 1; l <= m->Lh; ++l) gate_out.push_back({m->n, l});
  const Op hop{hid, m->HS}, nop{nodec, m->node_stride}, mop{msgc, 16};
  add_block({nop}, hid_out, SGG_PLAIN);
  for (int k = 0; k < L && !bad && !short_blob; ++k) {
    for (int i = 0; i < B; ++i) add_block(i == 0 ? std::vector<Op>{hop, hop, mop} : std::vector<Op>{hop}, gate_out, SGG_GATE);
    for (int i = 0; i < B; ++i) {
      const bool last = i == B - 1;
      add_block(i == 0 ? std::vector<Op>{hop, hop} : std::vector<Op>{hop}, last ? hid_out : gate_out, last ? SGG_PLAIN : SGG_GATE);
    }
  }
  for (int i = 0; i < B && !bad && !short_blob; ++i) add_block({hop}, gate_out, SGG_GATE);
  if (!bad && !short_blob) add_block({hop}, {{1, 1}}, SGG_OUTVEC);
  // norm parameters
  if (!short_blob && m->norm) {
    const int nw = m->n * (m->Lh + 1), nb0 = m->n;
    for (int k = 0; k < L; ++k) {
      const int sets = m->norm == 2 ? 2 : 1;
      for (int q = 0; q < sets; ++q) {
        if (p + nw + nb0 > pend) { short_blob = true; break; }
        size_t o1 = (host.size() + 63) & ~(size_t)63;
        host.resize(o1 + nw, 0.f);
        memcpy(host.data() + o1, p, sizeof(float) * nw);
        p += nw;
        size_t o2 = (host.size() + 63) & ~(size_t)63;
        host.resize(o2 + nb0, 0.f);
        memcpy(host.data() + o2, p, sizeof(float) * nb0);
        p += nb0;
        m->norm_off.push_back((int64_t)o1);
        m->norm_off.push_back((int64_t)o2);
      }
    }
  }
  if (short_blob || p != pend) {
    lb_sgg_destroy(m);
    return lb_fail(LB_ERR_ARG, "segnn weight blob has %lld floats, expected %s%lld", (long long)n_floats, short_blob ? "more than " : "",
                   (long long)(p - w));
  }
  size_t max_lds = 0;
  for (const sgg_block& b : m->blocks) max_lds = std::max(max_lds, b.lds_bytes);
  if (max_lds > 160 * 1024) {
    lb_sgg_destroy(m);
    return lb_fail(LB_ERR_UNSUPPORTED, "segnn: a tensor product needs %zu bytes of LDS (<= 160 KiB)", max_lds);
  }
  // the attribute belongs to the kernel, not to this model: only ever raise it (several models may be alive)
  static size_t lds_set[2] = {0, 0};
  const int which = m->Lh >= 2 ? 1 : 0;
  if (max_lds > lds_set[which] &&
      hipFuncSetAttribute(reinterpret_cast<const void*>(which ? k_sgg_tp<512> : k_sgg_tp<256>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)max_lds) != hipSuccess) {
    lb_sgg_destroy(m);
    return lb_fail(LB_ERR_HIP, "hipFuncSetAttribute(k_sgg_tp, %zu bytes of LDS) failed", max_lds);
  }
  lds_set[which] = std::max(lds_set[which], max_lds);
  int rc = sgg_alloc(&m->blob, host.size());
  if (!rc && hipMemcpy(m->blob, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess)
    rc = lb_fail(LB_ERR_HIP, "weight upload failed");
  if (!rc) rc = sgg_alloc(&m->ktab, ktab_host.size());
  if (!rc && hipMemcpy(m->ktab, ktab_host.data(), ktab_host.size() * sizeof(int32_t), hipMemcpyHostToDevice) != hipSuccess)
    rc = lb_fail(LB_ERR_HIP, "table upload failed");
  for (sgg_block& b : m->blocks) {
    b.a.ktab = m->ktab;
    b.a.w = m->blob;
    b.a.cg = m->blob;
    b.a.bias = m->blob + reinterpret_cast<size_t>(b.a.bias);
  }
  const int64_t BN = e->BN;
  if (!rc) rc = sgg_alloc(&m->xnode, (size_t)BN * 32);
  if (!rc) rc = sgg_alloc(&m->nodesv, (size_t)BN * m->node_stride);
  if (!rc) rc = sgg_alloc(&m->nattr4, (size_t)BN * 4);
  if (!rc && m->La >= 2) rc = sgg_alloc(&m->nattr, (size_t)BN * 12);
  for (float** q : {&m->f, &m->agg, &m->tn[0], &m->tn[1]})
    if (!rc) rc = sgg_alloc(q, (size_t)BN * m->HS);
  if (!rc && m->norm == 2) {
    rc = sgg_alloc(&m->bn_part, (size_t)e->g.B * m->bn_G * m->HS);
    for (float** q : {&m->bn_mean, &m->bn_scale, &m->bn_shift})
      if (!rc) rc = sgg_alloc(q, (size_t)e->g.B * m->HS);
  }
  if (rc) {
    lb_sgg_destroy(m);
    return rc;
  }
  *out = m;
  return LB_OK;
}

// SEGNN.__call__ (segnn.py:595-610) on the engine's current window + neighbor list -> e->acc
// lb_toc on every way out of a timed region (lbk_sgg_forward returns from between lb_tic and lb_toc on launch errors)
struct lb_toc_guard {
  lb_engine* e;
  bool armed = true;
  ~lb_toc_guard() { if (armed) lb_toc(e); }
  void done() { lb_toc(e); armed = false; }
};
int lbk_sgg_forward(lb_engine* e, lb_sgg* m) {
  hipStream_t s = e->stream;
  const int64_t BN = e->BN;
  const int B = m->desc.blocks_per_step, L = m->desc.num_mp_steps;
  LB_TRY(sgg_ensure_edges(m));
  const int64_t ecap = (int64_t)e->e_cap * e->g.B;
  lb_tic(e, LB_T_NODEFEAT);
  lb_toc_guard tg0{e};  // (an early return must not leave the timer open: ADVICE r05)
  LB_TRY(lbk_sg_prep(e, m->desc.homogeneous, m->desc.velocity_avg, m->node_ns4, m->node_nv4, m->xnode, m->eattr4, m->msgsv, m->nodesv,
                     m->nattr4, ecap));
  const float *eattr = m->eattr4, *nattr = m->nattr4;
  if (m->La >= 2) {
    hipLaunchKernelGGL(k_sgg_attr_edge, dim3((unsigned)((ecap + 255) / 256)), dim3(256), 0, s, e->ctrl, m->eattr4, m->eattr, ecap);
    hipLaunchKernelGGL(k_sgg_attr_node, dim3((unsigned)((BN + 255) / 256)), dim3(256), 0, s, e->ctrl, BN, e->g.isl - 1,
                       m->desc.velocity_avg, m->node_ns4, m->node_nv4, m->nodesv, e->row_ptr, m->eattr, m->nattr);
    eattr = m->eattr;
    nattr = m->nattr;
  }
  tg0.done();
  LB_HIP(hipGetLastError());
  auto tap = [&](int slot) -> int {
    if (m->tap)
      LB_HIP(hipMemcpyAsync(m->tap + (size_t)slot * BN * m->HS, m->f, sizeof(float) * BN * m->HS, hipMemcpyDeviceToDevice, s));
    return LB_OK;
  };
  size_t bi = 0;
  {
    const float* xs[1] = {m->nodesv};
    lb_tic(e, LB_T_ENC_NODE);
    lb_toc_guard tg1{e};  // (an early return must not leave the timer open: ADVICE r05)
    int rc = sgg_launch(m, m->blocks[bi++], BN, false, xs, nullptr, nattr, nullptr, m->f);
    tg1.done();
    if (rc) return rc;
  }
  LB_TRY(tap(0));
  const unsigned nb_seg = (unsigned)((BN * (m->HS / 4) + 255) / 256);
  for (int k = 0; k < L; ++k) {
    lb_tic(e, LB_T_EDGE_MLP);
    lb_toc_guard tg2{e};  // (an early return must not leave the timer open: ADVICE r05)
    float* cur = nullptr;
    for (int i = 0; i < B; ++i) {
      float* dst = m->te[i & 1];
      int rc;
      if (i == 0) {
        const float* xs[3] = {m->f, m->f, m->msgsv};
        const int32_t* ga[3] = {e->senders, e->receivers, nullptr};
        rc = sgg_launch(m, m->blocks[bi++], ecap, true, xs, ga, eattr, nullptr, dst);
      } else {
        const float* xs[1] = {cur};
        rc = sgg_launch(m, m->blocks[bi++], ecap, true, xs, nullptr, eattr, nullptr, dst);
      }
      if (rc) return rc;
      cur = dst;
    }
    if (m->norm == 2) LB_TRY(sgg_batch_norm(m, m->norm_off[(size_t)k * 4 + 0], m->norm_off[(size_t)k * 4 + 1], cur, true));
    tg2.done();
    lb_tic(e, LB_T_AGGREGATE);
    lb_toc_guard tg3{e};  // (an early return must not leave the timer open: ADVICE r05)
    hipLaunchKernelGGL(k_sgg_segsum, dim3(nb_seg), dim3(256), 0, s, e->ctrl, e->row_ptr, cur, m->agg, BN, m->HS / 4);
    tg3.done();
    lb_tic(e, LB_T_NODE_MLP);
    lb_toc_guard tg4{e};  // (an early return must not leave the timer open: ADVICE r05)
    const float* ncur = nullptr;
    for (int i = 0; i < B; ++i) {
      const bool last = i == B - 1;
      float* dst = last ? m->f : m->tn[i & 1];
      int rc;
      if (i == 0) {
        const float* xs[2] = {m->f, m->agg};
        rc = sgg_launch(m, m->blocks[bi++], BN, false, xs, nullptr, nattr, last ? m->f : nullptr, dst);
      } else {
        const float* xs[1] = {ncur};
        rc = sgg_launch(m, m->blocks[bi++], BN, false, xs, nullptr, nattr, last ? m->f : nullptr, dst);
      }
      if (rc) return rc;
      ncur = dst;
    }
    if (m->norm == 2) {
      LB_TRY(sgg_batch_norm(m, m->norm_off[(size_t)k * 4 + 2], m->norm_off[(size_t)k * 4 + 3], m->f, false));
    } else if (m->norm == 1) {
      sgg_bn_args a = sgg_bn_base(m, m->norm_off[(size_t)k * 2 + 0], m->norm_off[(size_t)k * 2 + 1], m->f, false);
      hipLaunchKernelGGL(k_sgg_inorm, dim3((unsigned)BN), dim3((unsigned)m->HS), 0, s, a, BN);
    }
    tg4.done();
    LB_TRY(tap(k + 1));
  }
  lb_tic(e, LB_T_DECODER);
  lb_toc_guard tg5{e};  // (an early return must not leave the timer open: ADVICE r05)
  const float* ncur = m->f;
  for (int i = 0; i < B; ++i) {
    const float* xs[1] = {ncur};
    LB_TRY(sgg_launch(m, m->blocks[bi++], BN, false, xs, nullptr, nattr, nullptr, m->tn[i & 1]));
    ncur = m->tn[i & 1];
  }
  {
    const float* xs[1] = {ncur};
    LB_TRY(sgg_launch(m, m->blocks[bi++], BN, false, xs, nullptr, nattr, nullptr, e->acc));
  }
  tg5.done();
  LB_HIP(hipGetLastError());
  return LB_OK;
}
