// lb_segnn.hip - SEGNN (steerable E(3)-equivariant GNN, lmax 1) forward pass on gfx950.
//
// Reference functions replaced (paths relative to the reference repo):
//   O3TensorProduct / O3TensorProductGate     lagrangebench/models/segnn.py:44-181
//   O3Embedding / O3Decoder                   lagrangebench/models/segnn.py:184-249
//   SEGNNLayer (message / update)             lagrangebench/models/segnn.py:252-362
//   SEGNN._transform / __call__               lagrangebench/models/segnn.py:513-610
// e3nn-jax conventions (tensor-product paths, Linear normalisation, gate) as restated in
// oracle/segnn_oracle.py (A1-A6) - parity is pinned to that restatement only.
//
// Data layout ("SV rows"): a feature row with ns scalars and nv vectors is stored
// [s(ns4) | vx(nv4) | vy(nv4) | vz(nv4)] fp32 with ns4/nv4 = counts padded to 4, i.e. vectors are
// component-major.  The hidden state 32x0e+32x1o is one 128-float (512 B) row.
//
// Kernel design: for lmax 1 the tensor product with the attribute (a0, a) followed by the
// e3nn Linear is four dense products sharing two weight matrices:
//     out_s    = [s a0 | (v.a)/sqrt3] Ws + b          (scalar channels)
//     out_v[c] = [s a_c |  v_c a0   ] Wv   c = x,y,z   (vector channels)
// One 256-thread block owns 32 rows; wave p computes part p (0 = scalars, 1..3 = x, y, z) with
// the transposed fp32 MFMA scheme of lb_gns.hip (weights = A operand in fragment order, the 32
// rows = B operand).  The attribute modulation is applied while the B operand is loaded, so the
// tensor-product intermediates never exist in memory.  The gate (silu on scalars, sigmoid gates
// on vectors) needs the scalar wave's gate channels in the vector waves: they sit at the same
// (lane, register) position of the accumulator, so one 4 KiB LDS hand-off does it.  Residual
// add, gating and the [s|vx|vy|vz] store are fused into the epilogue.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <utility>

#include "lb_device.h"

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

static constexpr float SG_Y0 = 0.28209479177387814f;      // 1 / (2 sqrt(pi))
static constexpr float SG_Y1 = 0.4886025119029199f;       // sqrt(3 / (4 pi))
static constexpr float SG_INV_SQRT3 = 0.5773502691896258f;
static constexpr float SG_C_SILU = 1.6765620f;            // 1/sqrt(E[silu(z)^2]),    z ~ N(0,1)
static constexpr float SG_C_SIGMOID = 1.8462292f;         // 1/sqrt(E[sigmoid(z)^2])

enum { SG_PLAIN = 0, SG_GATE = 1, SG_RESID = 2, SG_OUTVEC = 3 };

struct lb_sg_operand {
  const float* x;
  const int32_t* gather;  // row index per output row, or null (identity)
  int32_t stride, ns4, nv4;
};

struct lb_sg_tp_args {
  const lb_ctrl* ctrl;
  int64_t n_rows;          // static row count; ignored when rows_from_ctrl
  int32_t rows_from_ctrl;  // 1: rows = ctrl->n_edges_total
  int32_t n_op;
  lb_sg_operand op[3];
  int32_t seg_start[8];    // k offsets of the 2*n_op (scalar-derived, vector-derived) segments + end
  const float* attr;       // [rows][4]
  const float* ws;         // packed (8*nkq) x (32*ms_blocks), pre-scaled by 1/sqrt(K)
  const float* wv;         // packed (8*nkq) x 32
  const float* bias;       // [32*ms_blocks]
  int32_t nkq, ms_blocks;
  float* dst;              // PLAIN/GATE/RESID: [rows][128]; OUTVEC: [rows][4]
};

struct lb_sg_block {       // one O3TensorProduct(+Gate) on the device
  int32_t n_op, nkq, ms_blocks;
  int32_t ns4[3], nv4[3];
  int32_t seg_start[8];
  const float* ws;
  const float* wv;
  const float* bias;
};

struct lb_segnn {
  lb_segnn_desc desc;
  lb_engine* eng;
  lb_sgg* gen = nullptr;   // non-null: general irreps / norm (lb_segnn_gen.hip); nothing else below is used then
  int node_ns, node_nv, node_ns4, node_nv4, node_stride;
  float* blob;
  lb_sg_block embedding, output;
  std::vector<lb_sg_block> message, update, readout;  // [layer*B + i], readout[i]
  // node-sized scratch
  float* xnode;    // [BN][32]  engine feature row (GNS column order, no embedding)
  float* nodesv;   // [BN][node_stride]
  float* nattr;    // [BN][4]
  float* f;        // [BN][128] hidden state
  float* agg;      // [BN][128]; followed in the SAME allocation by
  float* part;     // [ceil(e_alloc / 16) + 2][2][128]: k_sg_msg's per-tile partial slots (one buffer descriptor)
  int64_t aggpart_bytes;
  float* tn[2];    // [BN][128] block intermediates
  // edge-sized scratch (regrown with the engine's e_alloc)
  int64_t e_alloc;
  float* eattr;    // [e_alloc][4]
  float* msgsv;    // [e_alloc][16]
  float* tap;
  std::vector<const float*> msg_image;  // per layer: LDS image of the fused message kernel
  std::vector<const float*> upd_image;  // per layer: LDS image of the fused update kernel
  const float* embed_image = nullptr;    // k_sg_embed (node prep + O3Embedding), k_sg_readout (O3Decoder + integrator)
  const float* readout_image = nullptr;
  bool fused_node = false;
  bool fused_msg;  // gather + both message blocks + segment_sum in one kernel (blocks_per_step == 2)
};

template <typename T>
static int sg_alloc(T** p, size_t n) {
  *p = nullptr;
  LB_HIP(hipMalloc((void**)p, (n ? n : 1) * sizeof(T)));
  return LB_OK;
}

// ------------------------------------------------------------------------------- attributes
// Edge attributes = spherical harmonics of the relative displacement, and the additional message
// features (rel_disp "1x1o", rel_dist "1x0e") as an SV row - segnn.py:556-560,577-584.
__global__ void k_sg_edge_prep(const lb_ctrl* __restrict__ ctrl, int dim,
                               const float* __restrict__ efeat, float* __restrict__ eattr,
                               float* __restrict__ msgsv, int64_t cap) {
  if (ctrl->overflow_step >= 0) return;
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= ctrl->n_edges_total || k >= cap) return;
  const f32x4 a = reinterpret_cast<const f32x4*>(efeat)[2 * k];
  const float dx = a[0], dy = a[1], dz = dim == 3 ? a[2] : 0.f, dist = dim == 3 ? a[3] : a[2];
  const float nrm = sqrtf(dx * dx + dy * dy + dz * dz);
  const float inv = nrm == 0.f ? 0.f : SG_Y1 / nrm;
  reinterpret_cast<f32x4*>(eattr)[k] = f32x4{SG_Y0, dx * inv, dy * inv, dz * inv};
  f32x4* m = reinterpret_cast<f32x4*>(msgsv) + 4 * k;
  m[0] = f32x4{dist, 0.f, 0.f, 0.f};
  m[1] = f32x4{dx, 0.f, 0.f, 0.f};
  m[2] = f32x4{dy, 0.f, 0.f, 0.f};
  m[3] = f32x4{dz, 0.f, 0.f, 0.f};
}

// Node input SV row + node attributes - segnn.py:536-575.  xnode is the engine's fp32 feature row
// [vel_hist(K*dim) | vel_mag(K) | bound(2 dim) | force(dim)].
__global__ void k_sg_node_prep(lb_geom g, int64_t BN, const lb_ctrl* __restrict__ ctrl,
                               const float* __restrict__ xnode, int xstride,
                               const int32_t* __restrict__ ptype, const int32_t* __restrict__ row_ptr,
                               const float* __restrict__ eattr, int homogeneous, int vel_avg,
                               int ns4, int nv4, float* __restrict__ nodesv,
                               float* __restrict__ nattr) {
  if (ctrl->overflow_step >= 0) return;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= BN) return;
  const int K = g.isl - 1, dim = g.dim;
  const float* x = xnode + i * xstride;
  float* o = nodesv + i * (ns4 + 3 * nv4);
  for (int j = 0; j < ns4 + 3 * nv4; ++j) o[j] = 0.f;
  int col = K * dim, s = 0, v = 0;
  float vm[3] = {0.f, 0.f, 0.f};
  for (int t = 0; t < K; ++t, ++v)
    for (int d = 0; d < dim; ++d) {
      const float val = x[t * dim + d];
      o[ns4 + d * nv4 + v] = val;
      if (vel_avg) vm[d] += val;
      else if (t == K - 1) vm[d] = val;
    }
  if (vel_avg && K > 1)
    for (int d = 0; d < 3; ++d) vm[d] = vm[d] / (float)K;
  if (g.has_vel_mag) {
    for (int t = 0; t < K; ++t) o[s++] = x[col + t];
    col += K;
  }
  if (g.has_bound) {
    for (int b = 0; b < 2; ++b, ++v)
      for (int d = 0; d < dim; ++d) o[ns4 + d * nv4 + v] = x[col + b * dim + d];
    col += 2 * dim;
  }
  if (g.force_kind != LB_FORCE_NONE) {
    for (int d = 0; d < dim; ++d) o[ns4 + d * nv4 + v] = x[col + d];
    ++v;
  }
  if (!homogeneous) {
    int t = ptype[i];
    if (t < 0) t += 9;
    for (int j = 0; j < 9; ++j) o[s + j] = (j == t) ? 1.f : 0.f;
  }
  // attributes: SH(velocity) + mean over incoming edges of SH(rel_disp); l = 0 entry forced to 1
  const float nrm = sqrtf(vm[0] * vm[0] + vm[1] * vm[1] + vm[2] * vm[2]);
  const float inv = nrm == 0.f ? 0.f : SG_Y1 / nrm;
  const int E = ctrl->n_edges_total;
  int k0 = row_ptr[i], k1 = row_ptr[i + 1];
  k0 = k0 < E ? k0 : E;
  k1 = k1 < E ? k1 : E;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int k = k0; k < k1; ++k) acc = acc + reinterpret_cast<const f32x4*>(eattr)[k];
  const float cnt = (float)((k1 - k0) > 1 ? (k1 - k0) : 1);
  reinterpret_cast<f32x4*>(nattr)[i] =
      f32x4{1.f, vm[0] * inv + acc[1] / cnt, vm[1] * inv + acc[2] / cnt, vm[2] * inv + acc[3] / cnt};
}

// SEGNN._transform for a caller outside this file (the training step, lb_train_segnn.h): engine features -> node SV rows,
// node / edge attributes, message SV rows.  xnode [BN][32], eattr [ecap][4], msgsv [ecap][16], nodesv [BN][ns4 + 3 nv4].
int lbk_sg_prep(lb_engine* e, int homogeneous, int vel_avg, int ns4, int nv4, float* xnode, float* eattr, float* msgsv,
                float* nodesv, float* nattr, int64_t ecap) {
  hipStream_t s = e->stream;
  LB_TRY(lbk_node_features_raw(e, xnode, 32));
  hipLaunchKernelGGL(k_sg_edge_prep, dim3((unsigned)((ecap + 255) / 256)), dim3(256), 0, s, e->ctrl, e->g.dim, e->efeat, eattr,
                     msgsv, ecap);
  hipLaunchKernelGGL(k_sg_node_prep, dim3((unsigned)((e->BN + 255) / 256)), dim3(256), 0, s, e->g, e->BN, e->ctrl, xnode, 32,
                     e->ptype, e->row_ptr, eattr, homogeneous, vel_avg, ns4, nv4, nodesv, nattr);
  LB_HIP(hipGetLastError());
  return LB_OK;
}

// -------------------------------------------------------------------- tensor product + linear
__device__ __forceinline__ float sg_silu(float x) { return x / (1.f + expf(-x)); }
__device__ __forceinline__ float sg_sigmoid(float x) { return 1.f / (1.f + expf(-x)); }

template <int MODE>
__global__ void __launch_bounds__(256) k_sg_tp(lb_sg_tp_args a) {
  if (a.ctrl->overflow_step >= 0) return;
  const int64_t n_rows = a.rows_from_ctrl ? (int64_t)a.ctrl->n_edges_total : a.n_rows;
  const int64_t row0 = (int64_t)blockIdx.x * 32;
  if (row0 >= n_rows) return;
  __shared__ float sgate[16 * 64];
  const int lane = threadIdx.x & 63, part = threadIdx.x >> 6;
  const int n = lane & 31, h = lane >> 5;
  const int64_t row = row0 + n;
  const bool valid = row < n_rows;
  const int64_t rl = valid ? row : n_rows - 1;
  const f32x4 at = reinterpret_cast<const f32x4*>(a.attr)[rl];
  const float amul = at[part];  // scalar segments: a0 (part 0) or a_c; vector segments use a0 / dot
  const float* base[3];
#pragma unroll
  for (int o = 0; o < 3; ++o) {
    base[o] = nullptr;
    if (o < a.n_op) {
      const int64_t idx = a.op[o].gather ? (int64_t)a.op[o].gather[rl] : rl;
      base[o] = a.op[o].x + idx * a.op[o].stride;
    }
  }
  const int nmb = part == 0 ? a.ms_blocks : 1;
  const f32x4* wp = reinterpret_cast<const f32x4*>(part == 0 ? a.ws : a.wv);
  f32x16 acc[2];
#pragma unroll
  for (int mb = 0; mb < 2; ++mb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[mb][r] = 0.f;

  if (nmb > 0) {
    for (int kq = 0; kq < a.nkq; ++kq) {
      const int k0 = 8 * kq + 4 * h;
      f32x4 x = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int o = 0; o < 3; ++o) {
        if (o < a.n_op) {
          const int s0 = a.seg_start[2 * o], s1 = a.seg_start[2 * o + 1], s2 = a.seg_start[2 * o + 2];
          if (k0 >= s0 && k0 < s1) {
            x = *reinterpret_cast<const f32x4*>(base[o] + (k0 - s0)) * amul;
          } else if (k0 >= s1 && k0 < s2) {
            const float* vb = base[o] + a.op[o].ns4 + (k0 - s1);
            const int nv4 = a.op[o].nv4;
            if (part == 0) {
              const f32x4 vx = *reinterpret_cast<const f32x4*>(vb);
              const f32x4 vy = *reinterpret_cast<const f32x4*>(vb + nv4);
              const f32x4 vz = *reinterpret_cast<const f32x4*>(vb + 2 * nv4);
              x = (vx * at[1] + vy * at[2] + vz * at[3]) * SG_INV_SQRT3;
            } else {
              x = *reinterpret_cast<const f32x4*>(vb + (part - 1) * nv4) * at[0];
            }
          }
        }
      }
      if (nmb == 2) {
        const f32x4 w0 = wp[(kq * 2 + 0) * 64 + lane];
        const f32x4 w1 = wp[(kq * 2 + 1) * 64 + lane];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc[0] = MFMA(w0[j], x[j], acc[0]);
          acc[1] = MFMA(w1[j], x[j], acc[1]);
        }
      } else {
        const f32x4 w0 = wp[kq * 64 + lane];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[0] = MFMA(w0[j], x[j], acc[0]);
      }
    }
  }
  // bias on the scalar channels (C layout: acc[mb][4q+j] <-> channel 32 mb + 8 q + 4 h + j)
  if (part == 0 && a.bias) {
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
      if (mb < nmb) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 b = reinterpret_cast<const f32x4*>(a.bias)[(32 * mb + 8 * q + 4 * h) / 4];
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[mb][4 * q + j] += b[j];
        }
      }
  }
  __syncthreads();  // every wave has read its operands: dst may alias them from here on

  if (MODE == SG_OUTVEC) {
    if (part > 0 && h == 0 && valid) a.dst[row * 4 + (part - 1)] = acc[0][0];
    return;
  }
  if (MODE == SG_GATE) {
    if (part == 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        sgate[r * 64 + lane] = SG_C_SIGMOID * sg_sigmoid(acc[1][r]);
        acc[0][r] = SG_C_SILU * sg_silu(acc[0][r]);
      }
    }
    __syncthreads();
    if (part > 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[0][r] *= sgate[r * 64 + lane];
    }
  }
  if (!valid) return;
  f32x4* d4 = reinterpret_cast<f32x4*>(a.dst + row * 128 + 32 * part);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    f32x4 o = {acc[0][4 * q], acc[0][4 * q + 1], acc[0][4 * q + 2], acc[0][4 * q + 3]};
    if (MODE == SG_RESID) o = o + d4[2 * q + h];
    d4[2 * q + h] = o;
  }
}

static int sg_launch(lb_engine* e, const lb_sg_block& b, int mode, int n_op_used,
                     const float* const* xs, const int32_t* const* gathers, const int32_t* strides,
                     const float* attr, bool edge_rows, float* dst) {
  lb_sg_tp_args a{};
  a.ctrl = e->ctrl;
  a.n_rows = e->BN;
  a.rows_from_ctrl = edge_rows ? 1 : 0;
  a.n_op = b.n_op;
  if (n_op_used != b.n_op) return lb_fail(LB_ERR_STATE, "segnn: operand count mismatch");
  for (int o = 0; o < b.n_op; ++o) {
    a.op[o].x = xs[o];
    a.op[o].gather = gathers ? gathers[o] : nullptr;
    a.op[o].stride = strides[o];
    a.op[o].ns4 = b.ns4[o];
    a.op[o].nv4 = b.nv4[o];
  }
  for (int i = 0; i < 8; ++i) a.seg_start[i] = b.seg_start[i];
  a.attr = attr;
  a.ws = b.ws;
  a.wv = b.wv;
  a.bias = b.ms_blocks > 0 ? b.bias : nullptr;
  a.nkq = b.nkq;
  a.ms_blocks = b.ms_blocks;
  a.dst = dst;
  const int64_t rows = edge_rows ? (int64_t)e->e_cap * e->g.B : e->BN;
  const int nb = (int)((rows + 31) / 32);
  switch (mode) {
    case SG_PLAIN: hipLaunchKernelGGL((k_sg_tp<SG_PLAIN>), dim3(nb), dim3(256), 0, e->stream, a); break;
    case SG_GATE: hipLaunchKernelGGL((k_sg_tp<SG_GATE>), dim3(nb), dim3(256), 0, e->stream, a); break;
    case SG_RESID: hipLaunchKernelGGL((k_sg_tp<SG_RESID>), dim3(nb), dim3(256), 0, e->stream, a); break;
    default: hipLaunchKernelGGL((k_sg_tp<SG_OUTVEC>), dim3(nb), dim3(256), 0, e->stream, a); break;
  }
  LB_HIP(hipGetLastError());
  return LB_OK;
}

// ------------------------------------------------------------------------------------ model
static int sg_ensure_edges(lb_segnn* m) {
  lb_engine* e = m->eng;
  if (m->e_alloc >= e->e_alloc && m->eattr) return LB_OK;
  LB_HIP(hipStreamSynchronize(e->stream));
  if (m->eattr) (void)hipFree(m->eattr);
  if (m->msgsv) (void)hipFree(m->msgsv);
  m->eattr = m->msgsv = nullptr;
  LB_TRY(sg_alloc(&m->eattr, (size_t)e->e_alloc * 4));
  LB_TRY(sg_alloc(&m->msgsv, (size_t)e->e_alloc * 16));
  if (m->agg) (void)hipFree(m->agg);
  m->agg = nullptr;
  const size_t n_agg = (size_t)e->BN * 128, n_part = ((size_t)e->e_alloc / 16 + 2) * 2 * 128;
  LB_TRY(sg_alloc(&m->agg, n_agg + n_part));
  m->part = m->agg + n_agg;
  m->aggpart_bytes = (int64_t)(n_agg + n_part) * 4;
  m->e_alloc = e->e_alloc;
  return LB_OK;
}

extern "C" void lb_segnn_destroy(lb_segnn* m) {
  if (!m) return;
  if (m->gen) lb_sgg_destroy(m->gen);
  for (void* b : {(void*)m->blob, (void*)m->xnode, (void*)m->nodesv, (void*)m->nattr, (void*)m->f,
                  (void*)m->agg, (void*)m->tn[0], (void*)m->tn[1], (void*)m->eattr, (void*)m->msgsv})
    if (b) (void)hipFree(b);
  delete m;
}

extern "C" int lb_segnn_create(lb_engine* e, const lb_segnn_desc* d, const float* w, int64_t n_floats,
                               lb_segnn** out) {
  if (!e || !d || !w || !out) return lb_fail(LB_ERR_ARG, "null argument");
  if (d->blocks_per_step < 1 || d->blocks_per_step > 8) return lb_fail(LB_ERR_ARG, "bad blocks_per_step");
  if (d->num_mp_steps < 0 || d->num_mp_steps > 64) return lb_fail(LB_ERR_ARG, "bad num_mp_steps");
  if (d->n_vels != e->g.isl - 1) return lb_fail(LB_ERR_ARG, "n_vels %d != input_seq_length-1", d->n_vels);
  if (d->hidden != 32 || d->lmax_hidden != 1 || d->lmax_attributes != 1 || d->norm != 0) {
    // not the configuration the fused kernels are built for: the general-irreps path
    lb_segnn* m = new lb_segnn();
    m->desc = *d;
    m->eng = e;
    m->blob = nullptr; m->xnode = m->nodesv = m->nattr = m->f = m->agg = m->part = nullptr;
    m->tn[0] = m->tn[1] = nullptr; m->eattr = m->msgsv = nullptr; m->tap = nullptr; m->e_alloc = 0;
    const int rc = lb_sgg_create(e, d, w, n_floats, &m->gen);
    if (rc) {
      delete m;
      return rc;
    }
    *out = m;
    return LB_OK;
  }
  const int C = 32, B = d->blocks_per_step, L = d->num_mp_steps, K = e->g.isl - 1;
  lb_segnn* m = new lb_segnn();
  m->desc = *d;
  m->eng = e;
  m->node_ns = (e->g.has_vel_mag ? K : 0) + (d->homogeneous ? 0 : 9);
  m->node_nv = K + (e->g.has_bound ? 2 : 0) + (e->g.force_kind != LB_FORCE_NONE ? 1 : 0);
  m->node_ns4 = (m->node_ns + 3) & ~3;
  m->node_nv4 = (m->node_nv + 3) & ~3;
  m->node_stride = m->node_ns4 + 3 * m->node_nv4;

  std::vector<float> host;
  auto put = [&](const float* src, size_t n) -> size_t {
    size_t off = (host.size() + 63) & ~(size_t)63;
    host.resize(off + n, 0.f);
    if (src) memcpy(host.data() + off, src, n * sizeof(float));
    return off;
  };
  struct Pending { lb_sg_block b; size_t ws, wv, bias; const float *raw_ws, *raw_wv, *raw_b; };
  std::vector<Pending> pend;
  const float* p = w;
  const float* pend_end = w + n_floats;
  bool short_blob = false;
  // operands: true (ns, nv) per operand; Ms / Mv true output widths
  auto add_block = [&](std::vector<std::pair<int, int>> ops, int Ms, int Mv) {
    Pending pd{};
    lb_sg_block& b = pd.b;
    b.n_op = (int)ops.size();
    int ktrue = 0, kp = 0;
    std::vector<int> rowmap;  // true row -> padded row
    for (int o = 0; o < b.n_op; ++o) {
      b.ns4[o] = (ops[o].first + 3) & ~3;
      b.nv4[o] = (ops[o].second + 3) & ~3;
      b.seg_start[2 * o] = kp;
      for (int j = 0; j < ops[o].first; ++j) rowmap.push_back(kp + j);
      kp += b.ns4[o];
      b.seg_start[2 * o + 1] = kp;
      for (int j = 0; j < ops[o].second; ++j) rowmap.push_back(kp + j);
      kp += b.nv4[o];
      ktrue += ops[o].first + ops[o].second;
    }
    for (int i = 2 * b.n_op; i < 8; ++i) b.seg_start[i] = kp;
    const int Kpad = (kp + 7) & ~7;
    b.nkq = Kpad / 8;
    b.ms_blocks = (Ms + 31) / 32;
    const float scale = 1.0f / sqrtf((float)ktrue);  // e3nn Linear, "element" normalisation (A4)
    const size_t need = (size_t)ktrue * Ms + (size_t)ktrue * Mv + Ms;
    if (p + need > pend_end) { short_blob = true; return; }
    auto pack = [&](const float* src, int M, int Mpad) -> size_t {
      std::vector<float> padded((size_t)Kpad * (M > 0 ? M : 1), 0.f), tmp((size_t)Kpad * Mpad, 0.f);
      for (int r = 0; r < ktrue; ++r)
        for (int c = 0; c < M; ++c) padded[(size_t)rowmap[r] * M + c] = src[(size_t)r * M + c] * scale;
      if (M > 0) lb_pack_weight(padded.data(), Kpad, M, Kpad, Mpad, tmp.data());
      return put(tmp.data(), tmp.size());
    };
    pd.raw_ws = p;
    pd.ws = pack(p, Ms, 32 * (b.ms_blocks > 0 ? b.ms_blocks : 1));
    p += (size_t)ktrue * Ms;
    pd.raw_wv = p;
    pd.wv = pack(p, Mv, 32);
    p += (size_t)ktrue * Mv;
    pd.raw_b = p;
    std::vector<float> bb(64, 0.f);
    memcpy(bb.data(), p, sizeof(float) * Ms);
    pd.bias = put(bb.data(), 64);
    p += Ms;
    pend.push_back(pd);
  };
  add_block({{m->node_ns, m->node_nv}}, C, C);
  for (int k = 0; k < L; ++k) {
    for (int i = 0; i < B; ++i) {
      if (i == 0) add_block({{C, C}, {C, C}, {1, 1}}, 2 * C, C);
      else add_block({{C, C}}, 2 * C, C);
    }
    for (int i = 0; i < B; ++i) {
      const bool last = i == B - 1;
      if (i == 0) add_block({{C, C}, {C, C}}, last ? C : 2 * C, C);
      else add_block({{C, C}}, last ? C : 2 * C, C);
    }
  }
  for (int i = 0; i < B; ++i) add_block({{C, C}}, 2 * C, C);
  add_block({{C, C}}, 0, 1);
  if (short_blob || p != pend_end) {
    delete m;
    return lb_fail(LB_ERR_ARG, "segnn weight blob has %lld floats, expected %lld", (long long)n_floats,
                   (long long)(p - w));
  }
  // fused message kernel (lb_segnn_msg.hip): one LDS image per layer; pend[1 + k*2B + {0,1}]
  std::vector<size_t> image_off, upd_image_off;
  if (B == 2) {
    std::vector<float> img((size_t)lb_sg_msg_image_floats());
    for (int k = 0; k < L; ++k) {
      const Pending& m0 = pend[1 + (size_t)k * 2 * B];
      const Pending& m1 = pend[2 + (size_t)k * 2 * B];
      lb_sg_msg_image(m0.raw_ws, m0.raw_wv, m0.raw_b, m1.raw_ws, m1.raw_wv, m1.raw_b, img.data());
      image_off.push_back(put(img.data(), img.size()));
    }
    std::vector<float> uimg((size_t)lb_sg_upd_image_floats());
    for (int k = 0; k < L; ++k) {
      const Pending& u0 = pend[3 + (size_t)k * 2 * B];
      const Pending& u1 = pend[4 + (size_t)k * 2 * B];
      lb_sg_upd_image(u0.raw_ws, u0.raw_wv, u0.raw_b, u1.raw_ws, u1.raw_wv, u1.raw_b, uimg.data());
      upd_image_off.push_back(put(uimg.data(), uimg.size()));
    }
  }
  size_t embed_off = 0, readout_off = 0;
  const bool node_ok = (B == 2) && m->node_ns <= 32 && m->node_nv <= 32;
  if (node_ok) {
    std::vector<float> ei((size_t)lb_sg_embed_image_floats());
    lb_sg_embed_image(pend[0].raw_ws, pend[0].raw_wv, pend[0].raw_b, m->node_ns, m->node_nv, ei.data());
    embed_off = put(ei.data(), ei.size());
    const Pending& r0 = pend[1 + (size_t)L * 2 * B];
    const Pending& r1 = pend[2 + (size_t)L * 2 * B];
    const Pending& ro = pend[3 + (size_t)L * 2 * B];
    std::vector<float> ri((size_t)lb_sg_readout_image_floats());
    lb_sg_readout_image(r0.raw_ws, r0.raw_wv, r0.raw_b, r1.raw_ws, r1.raw_wv, r1.raw_b, ro.raw_wv, ri.data());
    readout_off = put(ri.data(), ri.size());
  }
  int rc = sg_alloc(&m->blob, host.size());
  if (!rc && hipMemcpy(m->blob, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess)
    rc = lb_fail(LB_ERR_HIP, "weight upload failed");
  size_t idx = 0;
  auto take = [&]() {
    Pending& pd = pend[idx++];
    pd.b.ws = m->blob + pd.ws;
    pd.b.wv = m->blob + pd.wv;
    pd.b.bias = m->blob + pd.bias;
    return pd.b;
  };
  m->embedding = take();
  for (int k = 0; k < L; ++k) {
    for (int i = 0; i < B; ++i) m->message.push_back(take());
    for (int i = 0; i < B; ++i) m->update.push_back(take());
  }
  for (int i = 0; i < B; ++i) m->readout.push_back(take());
  m->output = take();
  for (size_t off : image_off) m->msg_image.push_back(m->blob + off);
  for (size_t off : upd_image_off) m->upd_image.push_back(m->blob + off);
  {
    const char* f = getenv("LB_SEGNN_FUSED");
    m->fused_msg = (B == 2) && !(f && f[0] == '0');
    // LB_SMALL_FUSED=0: node prep / embedding / readout / integrator as round 3's separate launches
    m->fused_node = m->fused_msg && node_ok && lb_fused_launches();
    if (node_ok) {
      m->embed_image = m->blob + embed_off;
      m->readout_image = m->blob + readout_off;
    }
  }
  const int64_t BN = e->BN;
  if (!rc) rc = sg_alloc(&m->xnode, (size_t)BN * 32);
  if (!rc) rc = sg_alloc(&m->nodesv, (size_t)BN * m->node_stride);
  if (!rc) rc = sg_alloc(&m->nattr, (size_t)BN * 4);
  if (!rc) rc = sg_alloc(&m->f, (size_t)BN * 128);
  if (!rc) rc = sg_alloc(&m->tn[0], (size_t)BN * 128);
  if (!rc) rc = sg_alloc(&m->tn[1], (size_t)BN * 128);
  if (rc) {
    lb_segnn_destroy(m);
    return rc;
  }
  *out = m;
  return LB_OK;
}

extern "C" int lb_segnn_set_tap(lb_segnn* m, float* tap) {
  if (!m) return lb_fail(LB_ERR_ARG, "null model");
  m->tap = tap;
  if (m->gen) lb_sgg_set_tap(m->gen, tap);
  return LB_OK;
}
extern "C" int32_t lb_segnn_row_floats(lb_segnn* m) { return !m ? 0 : m->gen ? lb_sgg_row_floats(m->gen) : 128; }

int lbk_segnn_forward(lb_engine* e, lb_segnn* m) {
  if (m->gen) return lbk_sgg_forward(e, m->gen);
  hipStream_t s = e->stream;
  const int64_t BN = e->BN;
  const int B = m->desc.blocks_per_step, L = m->desc.num_mp_steps;
  LB_TRY(sg_ensure_edges(m));
  const int nb_n = (int)((BN + 255) / 256);
  const int64_t ecap = (int64_t)e->e_cap * e->g.B;
  const int nb_e = (int)((ecap + 255) / 256);
  auto tap = [&](int slot) -> int {
    if (m->tap)
      LB_HIP(hipMemcpyAsync(m->tap + (size_t)slot * BN * 128, m->f, sizeof(float) * BN * 128,
                            hipMemcpyDeviceToDevice, s));
    return LB_OK;
  };

  lb_tic(e, LB_T_NODEFEAT);
  if (!(e->feat_done && e->feat_job.xnode == m->xnode))  // (rollout step: written by the neighbor-search launch)
    LB_TRY(lbk_node_features_raw(e, m->xnode, 32));
  if (m->fused_node) {
    lb_toc(e);
    lb_tic(e, LB_T_ENC_NODE);
    int rc = lbk_sg_embed(e, m->xnode, m->embed_image, m->f, m->nattr, m->desc.homogeneous, m->desc.velocity_avg);
    lb_toc(e);
    if (rc) return rc;
  } else {
  hipLaunchKernelGGL(k_sg_edge_prep, dim3(nb_e), dim3(256), 0, s, e->ctrl, e->g.dim, e->efeat,
                     m->eattr, m->msgsv, ecap);
  hipLaunchKernelGGL(k_sg_node_prep, dim3(nb_n), dim3(256), 0, s, e->g, BN, e->ctrl, m->xnode, 32,
                     e->ptype, e->row_ptr, m->eattr, m->desc.homogeneous, m->desc.velocity_avg,
                     m->node_ns4, m->node_nv4, m->nodesv, m->nattr);
  lb_toc(e);
  LB_HIP(hipGetLastError());
  }

  const int32_t s128[3] = {128, 128, 16};
  if (!m->fused_node) {
    const float* xs[1] = {m->nodesv};
    const int32_t st[1] = {m->node_stride};
    lb_tic(e, LB_T_ENC_NODE);
    int rc = sg_launch(e, m->embedding, SG_PLAIN, 1, xs, nullptr, st, m->nattr, false, m->f);
    lb_toc(e);
    if (rc) return rc;
  }
  LB_TRY(tap(0));
  float* eb[2] = {e->elat, e->msg};
  for (int k = 0; k < L; ++k) {
    // message: [f_sender | f_receiver | (rel_disp, rel_dist)] -> gated blocks (segnn.py:280-304)
    if (m->fused_msg) {
      lb_tic(e, LB_T_EDGE_MLP);
      int rc = lbk_sg_message(e, m->f, m->msg_image[k], m->agg, m->part, m->aggpart_bytes, false);
      lb_toc(e);
      if (rc) return rc;
    } else {
    lb_tic(e, LB_T_EDGE_MLP);
    const float* cur = nullptr;
    for (int i = 0; i < B; ++i) {
      float* dst = eb[i & 1];
      int rc;
      if (i == 0) {
        const float* xs[3] = {m->f, m->f, m->msgsv};
        const int32_t* ga[3] = {e->senders, e->receivers, nullptr};
        rc = sg_launch(e, m->message[k * B + i], SG_GATE, 3, xs, ga, s128, m->eattr, true, dst);
      } else {
        const float* xs[1] = {cur};
        rc = sg_launch(e, m->message[k * B + i], SG_GATE, 1, xs, nullptr, s128, m->eattr, true, dst);
      }
      if (rc) return rc;
      cur = dst;
    }
    lb_toc(e);
    lb_tic(e, LB_T_AGGREGATE);
    LB_TRY(lbk_segment_sum(e, cur, m->agg, 128));
    lb_toc(e);
    }
    // update: [f | agg] -> gated blocks -> linear block -> residual (segnn.py:306-334)
    lb_tic(e, LB_T_NODE_MLP);
    if (m->fused_msg) {
      int rc = lbk_sg_update(e, m->f, m->agg, m->part, m->nattr, m->upd_image[k], true);
      lb_toc(e);
      if (rc) return rc;
      LB_TRY(tap(k + 1));
      continue;
    }
    const float* ncur = nullptr;
    for (int i = 0; i < B; ++i) {
      const bool last = i == B - 1;
      float* dst = last ? m->f : m->tn[i & 1];
      int rc;
      if (i == 0) {
        const float* xs[2] = {m->f, m->agg};
        rc = sg_launch(e, m->update[k * B + i], last ? SG_RESID : SG_GATE, 2, xs, nullptr, s128, m->nattr, false, dst);
      } else {
        const float* xs[1] = {ncur};
        rc = sg_launch(e, m->update[k * B + i], last ? SG_RESID : SG_GATE, 1, xs, nullptr, s128, m->nattr, false, dst);
      }
      if (rc) return rc;
      ncur = dst;
    }
    lb_toc(e);
    LB_TRY(tap(k + 1));
  }
  lb_tic(e, LB_T_DECODER);
  if (m->fused_node) {
    int rc = lbk_sg_readout(e, m->f, m->nattr, m->readout_image, e->acc);
    lb_toc(e);
    return rc;
  }
  const float* ncur = m->f;
  for (int i = 0; i < B; ++i) {
    const float* xs[1] = {ncur};
    LB_TRY(sg_launch(e, m->readout[i], SG_GATE, 1, xs, nullptr, s128, m->nattr, false, m->tn[i & 1]));
    ncur = m->tn[i & 1];
  }
  {
    const float* xs[1] = {ncur};
    LB_TRY(sg_launch(e, m->output, SG_OUTVEC, 1, xs, nullptr, s128, m->nattr, false, e->acc));
  }
  lb_toc(e);
  return LB_OK;
}

__global__ void k_sg_acc_export(int64_t BN, int dim, const float* __restrict__ acc4,
                                float* __restrict__ out) {
  int64_t gi = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gi >= BN) return;
  for (int d = 0; d < dim; ++d) out[gi * dim + d] = acc4[gi * 4 + d];
}

extern "C" int lb_segnn_forward(lb_engine* e, lb_segnn* m, float* acc_out_dev) {
  if (!e || !m) return lb_fail(LB_ERR_ARG, "null argument");
  if (m->eng != e) return lb_fail(LB_ERR_ARG, "model was created for another engine");
  if (e->e_cap <= 0) return lb_fail(LB_ERR_STATE, "lb_segnn_forward before lb_nl_allocate");
  if (e->g.force_kind == LB_FORCE_BUFFER && !e->force)
    return lb_fail(LB_ERR_STATE, "LB_FORCE_BUFFER engine: call lb_set_force first");
  LB_TRY(lbk_segnn_forward(e, m));
  if (acc_out_dev) {
    const int nb = (int)((e->BN + 255) / 256);
    hipLaunchKernelGGL(k_sg_acc_export, dim3(nb), dim3(256), 0, e->stream, e->BN, e->g.dim, e->acc,
                       acc_out_dev);
    LB_HIP(hipGetLastError());
  }
  return LB_OK;
}

static int sg_forward_thunk(lb_engine* e, void* model) { return lbk_segnn_forward(e, (lb_segnn*)model); }

extern "C" int lb_segnn_rollout(lb_engine* e, lb_segnn* m, const double* traj_dev, int32_t T,
                                int32_t n_steps, double* pred_out_dev, int32_t* n_realloc_out) {
  if (!e || !m || !traj_dev || !pred_out_dev) return lb_fail(LB_ERR_ARG, "null argument");
  if (m->eng != e) return lb_fail(LB_ERR_ARG, "model was created for another engine");
  if (!m->gen) e->feat_job = lb_feat_job{m->xnode, nullptr, 0, 1, 32, e->ptype, e->force};  // node-feature rows ride along with the search
  const int rc = lb_rollout_generic(e, sg_forward_thunk, m, traj_dev, T, n_steps, pred_out_dev, n_realloc_out);
  e->feat_job = lb_feat_job{};
  return rc;
}
