// lb_train_segnn.h - the SEGNN training step (lmax 1) on the device; IMPLEMENTATION INCLUDE of lb_train.hip (one translation
// unit: it uses that file's fp32-MFMA products, ordered reductions, loss kernels and AdamW).  Round 5: SURVEY.md section 8f
// row N4 for config 5's model - the reference trainer is model-agnostic (train/trainer.py:35-89), round 4 could train GNS only.
//
// Reference: SEGNN.__call__ / SEGNNLayer / O3TensorProduct(Gate) of lagrangebench/models/segnn.py:44-362,595-610 under
// value_and_grad of _mse (train/trainer.py:35-60); e3nn-jax conventions as restated in oracle/segnn_oracle.py (A1 - A6:
// parity with e3nn-jax itself is unpinned), gradients checked against float64 torch autograd of oracle/segnn_torch.py.
//
// Design.  For lmax 1 an O3TensorProduct with the attribute (a0, a) followed by e3nn's Linear is four dense products that
// share two weight matrices (lb_segnn.hip):   out_s = [s a0 | (v.a)/sqrt3] Ws / sqrt K + b,   out_v[c] = [s a_c | v_c a0] Wv / sqrt K.
// The training step makes that literal.  Per block and row r the modulated inputs are FOUR rows 4 r + p of a matrix Z
// (p = 0: the scalar part, p = 1..3: the x / y / z part; K columns zero-padded to a multiple of 16, pre-multiplied by
// 1 / sqrt K) and both weight matrices sit side by side in ONE 128-column operand W = [Ws (cols 0 .. Ms-1) | 0 | Wv (cols
// 64 .. 64+Mv-1) | 0]:   raw = Z W  (4 R x 128) holds every output the block needs - row 4 r: the scalar outputs in columns
// < Ms, rows 4 r + 1 + c: the vector outputs in columns 64 .. - plus cross terms nobody reads (Ws on vector rows and vice
// versa: 2x the necessary flops, on a row next to the hot path).  What that buys: every product of the forward AND the backward
// is a 128-wide tall-skinny fp32 GEMM that k_lin32f / k_dw_part already run (dW = Z^T d raw gives [dWs | dWv] in one ordered
// reduction because d raw is written with zeros in the cross-term columns; dZ = d raw W^T), and the SEGNN-specific work is four
// small elementwise kernels: k_sgt_in (operands (+ gathers) x attribute -> Z), k_sgt_out (bias, gate / residual -> SV rows),
// and their transposes.  Z and raw of every block are kept for the backward (~1 GB per 10 k edges of a 10-layer network).
// Gathers (f[snd], f[rcv]) are transposed without atomics: per-edge gradient rows, then k_sgt_scatter adds, per node, the
// rows of the edges it sends (sender-sorted permutation, ascending edge index) and receives (CSR row) in a fixed order -
// a step is bit-reproducible.  jraph.segment_sum's transpose is a gather folded into k_sgt_out_bwd's load.
// Limits: hidden multiplicity <= 32 (scalar_units 64, every shipped config), lmax 1, norm None - as the inference path.
// SV rows: [s (32) | vx (32) | vy (32) | vz (32)] fp32, zero beyond the real counts.
#pragma once

enum { SGT_PLAIN = 0, SGT_GATE = 1, SGT_OUTVEC = 2 };

struct lb_sgt_block {   // one O3TensorProduct(+Gate)
  int K, Kp, Ms, Mv, mode;
  int n_op, ns[3], nv[3];      // operands: real scalar / vector counts
  int64_t off_w, off_b;        // float offsets into the device blobs: W (Kp x 128), b (128)
  bool edge;                   // rows = E (message blocks) or BN
  float *Z = nullptr, *raw = nullptr;  // saved for the backward: (4 R x Kp), (4 R x 128)
};

struct lb_sgt_op {
  const float* x;        // SV rows
  const int32_t* gidx;   // row index per output row, or null
  float* dx;             // backward: where this operand's gradient rows go (null: no gradient needed)
  int stride, ns, nv, ns4, nv4;
  int accum;             // backward: dx += instead of =
};
struct lb_sgt_in_args {
  lb_sgt_op op[3];
  int n_op, K, Kp;
  int kstart[4];
  const float* attr;     // [R][4]
  float scale;
  int64_t R;
  float* Z;              // forward: out; backward: dZ in
};

struct lb_sgt {
  lb_segnn_desc desc;
  int node_ns, node_nv, node_ns4, node_nv4, node_stride;
  std::vector<lb_sgt_block> blocks;  // call order: embedding; per layer message_0.., update_0..; readout_0..; output
  int64_t cap_n = 0, cap_e = 0;
  float *xnode = nullptr, *nodesv = nullptr, *nattr = nullptr, *eattr = nullptr, *msgsv = nullptr;
  float *f = nullptr, *agg = nullptr, *tn[2] = {nullptr, nullptr}, *te[2] = {nullptr, nullptr};  // SV rows (BN / E x 128)
  float *df = nullptr, *dagg = nullptr, *dtn[2] = {nullptr, nullptr}, *dte[2] = {nullptr, nullptr};
  float *dFs = nullptr, *dFr = nullptr;   // per-edge gradient rows of f[snd], f[rcv]
  float *draw = nullptr, *dZ = nullptr;   // (4 Rmax x 128), (4 Rmax x 256)
};

// ------------------------------------------------------------------------------------------------------- kernels
__device__ __forceinline__ float sgt_sigmoid(float x) { return 1.f / (1.f + expf(-x)); }
static constexpr float SGT_INV_SQRT3 = 0.5773502691896258f;
static constexpr float SGT_C_SILU = 1.6765620f;     // 1 / sqrt(E[silu(z)^2]),    z ~ N(0, 1)  (oracle/segnn_oracle.py A5)
static constexpr float SGT_C_SIGMOID = 1.8462292f;  // 1 / sqrt(E[sigmoid(z)^2])

// Z[4 r + p][k] for k < Kp: the tensor product of the operands with the attribute (segnn_oracle.tp_inputs), x 1 / sqrt K
__global__ void k_sgt_in(lb_sgt_in_args a) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.R * a.Kp) return;
  const int64_t r = i / a.Kp;
  const int k = (int)(i % a.Kp);
  float z0 = 0.f, z1 = 0.f, z2 = 0.f, z3 = 0.f;
  if (k < a.K) {
    const f32x4 at = reinterpret_cast<const f32x4*>(a.attr)[r];
    int o = 0;
    while (o + 1 < a.n_op && k >= a.kstart[o + 1]) ++o;
    const lb_sgt_op& op = a.op[o];
    const int c = k - a.kstart[o];
    const int64_t row = op.gidx ? (int64_t)op.gidx[r] : r;
    const float* x = op.x + row * op.stride;
    if (c < op.ns) {
      const float s = x[c];
      z0 = s * at[0]; z1 = s * at[1]; z2 = s * at[2]; z3 = s * at[3];
    } else {
      const int j = c - op.ns;
      const float vx = x[op.ns4 + j], vy = x[op.ns4 + op.nv4 + j], vz = x[op.ns4 + 2 * op.nv4 + j];
      z0 = ((vx * at[1] + vy * at[2]) + vz * at[3]) * SGT_INV_SQRT3;
      z1 = vx * at[0]; z2 = vy * at[0]; z3 = vz * at[0];
    }
    z0 *= a.scale; z1 *= a.scale; z2 *= a.scale; z3 *= a.scale;
  }
  float* z = a.Z + (4 * r) * a.Kp + k;
  z[0] = z0;
  z[a.Kp] = z1;
  z[2 * (int64_t)a.Kp] = z2;
  z[3 * (int64_t)a.Kp] = z3;
}
// its transpose: dZ (4 R x Kp) -> the operands' gradient rows (SV layout 32 | 32 | 32 | 32).  Thread (r, j < 32) owns scalar j
// and vector j of every operand.
__global__ void k_sgt_in_bwd(lb_sgt_in_args a) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.R * 32) return;
  const int64_t r = i / 32;
  const int j = (int)(i % 32);
  const f32x4 at = reinterpret_cast<const f32x4*>(a.attr)[r];
  const float* z = a.Z + (4 * r) * a.Kp;
  const int64_t kp = a.Kp;
  for (int o = 0; o < a.n_op; ++o) {
    const lb_sgt_op& op = a.op[o];
    if (!op.dx) continue;
    float* d = op.dx + r * 128;
    float ds = 0.f, dvx = 0.f, dvy = 0.f, dvz = 0.f;
    if (j < op.ns) {
      const int k = a.kstart[o] + j;
      ds = (((z[k] * at[0] + z[kp + k] * at[1]) + z[2 * kp + k] * at[2]) + z[3 * kp + k] * at[3]) * a.scale;
    }
    if (j < op.nv) {
      const int k = a.kstart[o] + op.ns + j;
      const float g0 = z[k] * SGT_INV_SQRT3;
      dvx = (g0 * at[1] + z[kp + k] * at[0]) * a.scale;
      dvy = (g0 * at[2] + z[2 * kp + k] * at[0]) * a.scale;
      dvz = (g0 * at[3] + z[3 * kp + k] * at[0]) * a.scale;
    }
    if (op.accum) {
      d[j] += ds; d[32 + j] += dvx; d[64 + j] += dvy; d[96 + j] += dvz;
    } else {
      d[j] = ds; d[32 + j] = dvx; d[64 + j] = dvy; d[96 + j] = dvz;
    }
  }
}

struct lb_sgt_out_args {
  const float* raw;      // (4 R x 128)
  const float* bias;     // [128]: entries < Ms
  const float* resid;    // forward PLAIN: SV rows added to the result, or null
  float* out;            // forward: SV rows (R x 128); OUTVEC: (R x dim)
  const float* dout;     // backward: gradient of `out` (SV rows; OUTVEC: (R x dim)), row gidx[r] if gidx
  const int32_t* gidx;
  float* draw;           // backward: (4 R x 128)
  int64_t R;
  int Ms, Mv, mode, dim;
};
// raw -> the block's output: bias, gate (e3nn.gate: silu on the first Ms - Mv scalars, the last Mv scalars gate the vectors
// through a sigmoid, both with their second-moment constants) or nothing (+ residual); thread (r, j < 32)
__global__ void k_sgt_out(lb_sgt_out_args a) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.R * 32) return;
  const int64_t r = i / 32;
  const int j = (int)(i % 32);
  const float* w = a.raw + (4 * r) * 128;
  if (a.mode == SGT_OUTVEC) {
    if (j < a.dim) a.out[r * a.dim + j] = w[(1 + j) * 128 + 64];
    return;
  }
  float s = 0.f, vx = 0.f, vy = 0.f, vz = 0.f;
  const int n_act = a.mode == SGT_GATE ? a.Ms - a.Mv : a.Ms;
  if (j < n_act) {
    const float z = w[j] + a.bias[j];
    s = a.mode == SGT_GATE ? SGT_C_SILU * (z * sgt_sigmoid(z)) : z;
  }
  if (j < a.Mv) {
    float g = 1.f;
    if (a.mode == SGT_GATE) g = SGT_C_SIGMOID * sgt_sigmoid(w[n_act + j] + a.bias[n_act + j]);
    vx = w[128 + 64 + j] * g; vy = w[256 + 64 + j] * g; vz = w[384 + 64 + j] * g;
  }
  float* o = a.out + r * 128;
  if (a.resid) {
    const float* q = a.resid + r * 128;
    s += q[j]; vx += q[32 + j]; vy += q[64 + j]; vz += q[96 + j];
  }
  o[j] = s; o[32 + j] = vx; o[64 + j] = vy; o[96 + j] = vz;
}
// its transpose: d out -> d raw with zeros in every column the forward does not read.  Thread (r, j < 32) owns the columns
// {j, 32 + j, 64 + j, 96 + j} of the four rows.
__global__ void k_sgt_out_bwd(lb_sgt_out_args a) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.R * 32) return;
  const int64_t r = i / 32;
  const int j = (int)(i % 32);
  const float* w = a.raw + (4 * r) * 128;
  float* d = a.draw + (4 * r) * 128;
  const int64_t gr = a.gidx ? (int64_t)a.gidx[r] : r;
  float r0[2] = {0.f, 0.f}, rv[3] = {0.f, 0.f, 0.f};   // row 0: columns j, 32 + j; rows 1 + c: column 64 + j
  if (a.mode == SGT_OUTVEC) {
    if (j == 0)
      for (int c = 0; c < a.dim; ++c) rv[c] = a.dout[gr * a.dim + c];
  } else {
    const float* g = a.dout + gr * 128;
    const int n_act = a.mode == SGT_GATE ? a.Ms - a.Mv : a.Ms;
    for (int h = 0; h < 2; ++h) {
      const int col = 32 * h + j;
      if (col < n_act) {  // an activated (or plain) scalar: its gradient is channel `col` of the SV row (n_act <= 32)
        const float gs = g[col];
        if (a.mode == SGT_GATE) {
          const float z = w[col] + a.bias[col], sg = sgt_sigmoid(z);
          r0[h] = SGT_C_SILU * (sg * (1.f + z * (1.f - sg))) * gs;
        } else {
          r0[h] = gs;
        }
      } else if (a.mode == SGT_GATE && col < a.Ms) {  // the gate of vector channel q
        const int q = col - n_act;
        const float sg = sgt_sigmoid(w[col] + a.bias[col]);
        const float dg = (g[32 + q] * w[128 + 64 + q] + g[64 + q] * w[256 + 64 + q]) + g[96 + q] * w[384 + 64 + q];
        r0[h] = SGT_C_SIGMOID * (sg * (1.f - sg)) * dg;
      }
    }
    if (j < a.Mv) {
      float gt = 1.f;
      if (a.mode == SGT_GATE) gt = SGT_C_SIGMOID * sgt_sigmoid(w[n_act + j] + a.bias[n_act + j]);
      rv[0] = g[32 + j] * gt; rv[1] = g[64 + j] * gt; rv[2] = g[96 + j] * gt;
    }
  }
  d[j] = r0[0]; d[32 + j] = r0[1]; d[64 + j] = 0.f; d[96 + j] = 0.f;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float* dc = d + (1 + c) * 128;
    dc[j] = 0.f; dc[32 + j] = 0.f; dc[64 + j] = rv[c]; dc[96 + j] = 0.f;
  }
}
// transpose of the two gathers of the message input: dn[i] += sum over the edges i SENDS of A[e] (sender-sorted permutation,
// ascending edge index) + sum over the edges i RECEIVES of Bm[e] (its CSR row).  One 32-lane group per node, no atomics.
__global__ void k_sgt_scatter(const float* __restrict__ A, const float* __restrict__ Bm, const int32_t* __restrict__ snd_ptr,
                              const int32_t* __restrict__ snd_perm, const int32_t* __restrict__ row_ptr, float* __restrict__ dn,
                              int64_t N, int64_t E) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * 32) return;
  const int64_t r = i / 32;
  const int q = (int)(i % 32);
  f32x4 as = {0.f, 0.f, 0.f, 0.f}, ar = {0.f, 0.f, 0.f, 0.f};
  const int s0 = snd_ptr[r], s1 = snd_ptr[r + 1];
  for (int j = s0; j < s1; ++j) as = as + reinterpret_cast<const f32x4*>(A)[(int64_t)snd_perm[j] * 32 + q];
  int k0 = row_ptr[r], k1 = row_ptr[r + 1];
  k0 = k0 < E ? k0 : (int)E;
  k1 = k1 < E ? k1 : (int)E;
  for (int k = k0; k < k1; ++k) ar = ar + reinterpret_cast<const f32x4*>(Bm)[(int64_t)k * 32 + q];
  reinterpret_cast<f32x4*>(dn)[i] = reinterpret_cast<f32x4*>(dn)[i] + (as + ar);
}

// ------------------------------------------------------------------------------------------------------- host
static void sgt_free(lb_gns_train* t) {
  lb_sgt* g = t->sg;
  if (!g) return;
  std::vector<void*> bufs = {g->xnode, g->nodesv, g->nattr, g->eattr, g->msgsv, g->f, g->agg, g->tn[0], g->tn[1], g->te[0], g->te[1],
                             g->df, g->dagg, g->dtn[0], g->dtn[1], g->dte[0], g->dte[1], g->dFs, g->dFr, g->draw, g->dZ};
  for (lb_sgt_block& b : g->blocks) {
    bufs.push_back(b.Z);
    bufs.push_back(b.raw);
  }
  for (void* p : bufs)
    if (p) (void)hipFree(p);
  delete g;
  t->sg = nullptr;
}

static int sgt_ensure(lb_gns_train* t, int64_t BN, int64_t E) {
  lb_sgt* g = t->sg;
  if (BN <= g->cap_n && E <= g->cap_e && g->f) return LB_OK;
  LB_HIP(hipStreamSynchronize(t->eng->stream));
  const int64_t cn = std::max(BN, g->cap_n), ce = std::max<int64_t>(E + E / 8 + 1024, g->cap_e), cm = std::max(cn, ce);
  LB_TRY(tr_alloc(&g->xnode, (size_t)cn * 32));
  LB_TRY(tr_alloc(&g->nodesv, (size_t)cn * g->node_stride));
  LB_TRY(tr_alloc(&g->nattr, (size_t)cn * 4));
  LB_TRY(tr_alloc(&g->eattr, (size_t)ce * 4));
  LB_TRY(tr_alloc(&g->msgsv, (size_t)ce * 16));
  for (float** p : {&g->f, &g->agg, &g->tn[0], &g->tn[1], &g->df, &g->dagg, &g->dtn[0], &g->dtn[1]}) LB_TRY(tr_alloc(p, (size_t)cn * 128));
  for (float** p : {&g->te[0], &g->te[1], &g->dte[0], &g->dte[1], &g->dFs, &g->dFr}) LB_TRY(tr_alloc(p, (size_t)ce * 128));
  LB_TRY(tr_alloc(&g->draw, (size_t)4 * cm * 128));
  LB_TRY(tr_alloc(&g->dZ, (size_t)4 * cm * 256));
  int64_t red = 4096;
  for (lb_sgt_block& b : g->blocks) {
    const int64_t R = b.edge ? ce : cn;
    LB_TRY(tr_alloc(&b.Z, (size_t)4 * R * b.Kp));
    LB_TRY(tr_alloc(&b.raw, (size_t)4 * R * 128));
    red += (dw_groups_max(4 * R) * (b.Kp + 1) * 128 + 63) / 64 * 64 + (4 * R + 127) / 128 * 128 + 64;
  }
  // the training core's scratch (train_ensure's part that both models need)
  t->red_cap = red;
  LB_TRY(tr_alloc(&t->dwpart, (size_t)t->red_cap));
  if (!t->red_dev) {
    LB_TRY(lb_alloc(&t->red_dev, (size_t)LB_RED_MAX));
    LB_HIP(hipHostMalloc((void**)&t->red_host, sizeof(lb_red_ent) * (LB_RED_MAX + 1)));   // (+ the step's status words)
  }
  LB_TRY(tr_alloc(&t->pred, (size_t)cn * 4));
  LB_TRY(tr_alloc(&t->dy, (size_t)cn * 4));
  LB_TRY(tr_alloc(&t->node_w, (size_t)cn));
  LB_TRY(tr_alloc(&t->loss_part, (size_t)(cn / 64 + 8)));
  g->cap_n = cn;
  g->cap_e = ce;
  return LB_OK;
}

// one block forward: operands -> Z -> raw = Z W -> out
static int sgt_fwd(lb_gns_train* t, lb_sgt_block& b, int64_t R, const lb_sgt_op* ops, const float* attr, const float* resid,
                   float* out) {
  if (R == 0) return LB_OK;
  hipStream_t s = t->eng->stream;
  lb_sgt_in_args a{};
  a.n_op = b.n_op; a.K = b.K; a.Kp = b.Kp; a.attr = attr; a.scale = 1.0f / sqrtf((float)b.K); a.R = R; a.Z = b.Z;
  int k = 0;
  for (int o = 0; o < b.n_op; ++o) {
    a.op[o] = ops[o];
    a.kstart[o] = k;
    k += b.ns[o] + b.nv[o];
  }
  a.kstart[b.n_op] = k;
  hipLaunchKernelGGL(k_sgt_in, GRID1(R * b.Kp), 0, s, a);
  LB_TRY(gemm_nn(t, 4 * R, 128, b.Kp, b.Z, b.Kp, t->w + b.off_w, b.raw, 128));
  lb_sgt_out_args oa{};
  oa.raw = b.raw; oa.bias = t->w + b.off_b; oa.resid = resid; oa.out = out; oa.R = R; oa.Ms = b.Ms; oa.Mv = b.Mv; oa.mode = b.mode;
  oa.dim = t->eng->g.dim;
  hipLaunchKernelGGL(k_sgt_out, GRID1(R * 32), 0, s, oa);
  return LB_OK;
}
// one block backward: d out (rows gidx[r] of dout if gidx) -> parameter gradients (+=) and the operands' gradient rows
static int sgt_bwd(lb_gns_train* t, lb_sgt_block& b, int64_t R, const float* dout, const int32_t* gidx, lb_sgt_op* ops,
                   const float* attr) {
  if (R == 0) return LB_OK;
  hipStream_t s = t->eng->stream;
  lb_sgt* g = t->sg;
  lb_sgt_out_args oa{};
  oa.raw = b.raw; oa.bias = t->w + b.off_b; oa.dout = dout; oa.gidx = gidx; oa.draw = g->draw; oa.R = R; oa.Ms = b.Ms; oa.Mv = b.Mv;
  oa.mode = b.mode; oa.dim = t->eng->g.dim;
  hipLaunchKernelGGL(k_sgt_out_bwd, GRID1(R * 32), 0, s, oa);
  // dW = Z^T d raw; db = the column sums of d raw's first Ms columns (the vector rows hold zeros there), which k_dw_part forms
  // on the way: only those Ms of its 128 sums are added to the bias gradient, the padded entries stay exactly zero
  if (!dw_acc(t, 4 * R, b.Kp, b.Z, b.Kp, g->draw, t->g + b.off_w, b.Ms ? t->g + b.off_b : nullptr, b.Ms))
    return LB_ERR_STATE;  // (red_slot said why)
  bool need = false;
  for (int o = 0; o < b.n_op; ++o) need = need || ops[o].dx;
  if (!need) return LB_OK;
  LB_TRY(gemm_nt(t, 4 * R, 128, b.Kp, g->draw, t->w + b.off_w, g->dZ, b.Kp));
  lb_sgt_in_args a{};
  a.n_op = b.n_op; a.K = b.K; a.Kp = b.Kp; a.attr = attr; a.scale = 1.0f / sqrtf((float)b.K); a.R = R; a.Z = g->dZ;
  int k = 0;
  for (int o = 0; o < b.n_op; ++o) {
    a.op[o] = ops[o];
    a.kstart[o] = k;
    k += b.ns[o] + b.nv[o];
  }
  a.kstart[b.n_op] = k;
  hipLaunchKernelGGL(k_sgt_in_bwd, GRID1(R * 32), 0, s, a);
  return LB_OK;
}

static lb_sgt_op sgt_op(const float* x, int ns, int nv, float* dx = nullptr, int accum = 0, const int32_t* gidx = nullptr,
                        int stride = 128, int ns4 = 32, int nv4 = 32) {
  lb_sgt_op o{};
  o.x = x; o.gidx = gidx; o.dx = dx; o.stride = stride; o.ns = ns; o.nv = nv; o.ns4 = ns4; o.nv4 = nv4; o.accum = accum;
  return o;
}

extern "C" int lb_segnn_train_create(lb_engine* e, const lb_segnn_desc* d, const float* w, int64_t n_floats, lb_gns_train** out) {
  if (!e || !d || !w || !out) return lb_fail(LB_ERR_ARG, "null argument");
  if (d->hidden < 1 || d->hidden > 32)
    return lb_fail(LB_ERR_UNSUPPORTED, "segnn training: hidden multiplicity %d not built (<= 32: scalar_units <= 64, lmax 1)", d->hidden);
  if (d->lmax_hidden != 1 || d->lmax_attributes != 1 || d->norm != 0)
    return lb_fail(LB_ERR_UNSUPPORTED, "segnn training: lmax_hidden %d / lmax_attributes %d / norm %d not built (1 / 1 / none; inference has them)",
                   d->lmax_hidden, d->lmax_attributes, d->norm);
  if (d->blocks_per_step < 1 || d->blocks_per_step > 8) return lb_fail(LB_ERR_ARG, "bad blocks_per_step");
  if (d->num_mp_steps < 0 || d->num_mp_steps > 64) return lb_fail(LB_ERR_ARG, "bad num_mp_steps");
  if (d->n_vels != e->g.isl - 1) return lb_fail(LB_ERR_ARG, "n_vels %d != input_seq_length-1", d->n_vels);
  const int C = d->hidden, B = d->blocks_per_step, L = d->num_mp_steps, Kv = e->g.isl - 1;
  lb_gns_train* t = new lb_gns_train();
  lb_sgt* g = new lb_sgt();
  t->sg = g;
  t->eng = e;
  g->desc = *d;
  g->node_ns = (e->g.has_vel_mag ? Kv : 0) + (d->homogeneous ? 0 : 9);
  g->node_nv = Kv + (e->g.has_bound ? 2 : 0) + (e->g.force_kind != LB_FORCE_NONE ? 1 : 0);
  g->node_ns4 = (g->node_ns + 3) & ~3;
  g->node_nv4 = (g->node_nv + 3) & ~3;
  g->node_stride = g->node_ns4 + 3 * g->node_nv4;
  if (g->node_ns > 32 || g->node_nv > 32) {
    lb_gns_train_destroy(t);
    return lb_fail(LB_ERR_UNSUPPORTED, "segnn training: %d scalar / %d vector node features (<= 32 each)", g->node_ns, g->node_nv);
  }
  int64_t o = 0, oc = 0;
  auto add = [&](std::vector<std::pair<int, int>> ops, int Ms, int Mv, int mode, bool edge) {
    lb_sgt_block b{};
    b.n_op = (int)ops.size();
    for (int q = 0; q < b.n_op; ++q) {
      b.ns[q] = ops[q].first;
      b.nv[q] = ops[q].second;
      b.K += ops[q].first + ops[q].second;
    }
    b.Kp = (b.K + 15) / 16 * 16;   // (128 / 256 at first: the blocks with 64 and 130 channels did twice the work)
    b.Ms = Ms; b.Mv = Mv; b.mode = mode; b.edge = edge;
    b.off_w = o; o += (int64_t)b.Kp * 128;
    b.off_b = o; o += 128;
    // caller's blob (SEGNN.flatten): ws (K x Ms), wv (K x Mv), b (Ms)
    for (int k = 0; k < b.K; ++k)
      for (int m = 0; m < Ms; ++m) t->cmap.push_back(b.off_w + (int64_t)k * 128 + m);
    for (int k = 0; k < b.K; ++k)
      for (int m = 0; m < Mv; ++m) t->cmap.push_back(b.off_w + (int64_t)k * 128 + 64 + m);
    for (int m = 0; m < Ms; ++m) t->cmap.push_back(b.off_b + m);
    oc += (int64_t)b.K * (Ms + Mv) + Ms;
    g->blocks.push_back(b);
  };
  add({{g->node_ns, g->node_nv}}, C, C, SGT_PLAIN, false);
  for (int k = 0; k < L; ++k) {
    for (int i = 0; i < B; ++i) {
      if (i == 0) add({{C, C}, {C, C}, {1, 1}}, 2 * C, C, SGT_GATE, true);
      else add({{C, C}}, 2 * C, C, SGT_GATE, true);
    }
    for (int i = 0; i < B; ++i) {
      const bool last = i == B - 1;
      if (i == 0) add({{C, C}, {C, C}}, last ? C : 2 * C, C, last ? SGT_PLAIN : SGT_GATE, false);
      else add({{C, C}}, last ? C : 2 * C, C, last ? SGT_PLAIN : SGT_GATE, false);
    }
  }
  for (int i = 0; i < B; ++i) add({{C, C}}, 2 * C, C, SGT_GATE, false);
  add({{C, C}}, 0, 1, SGT_OUTVEC, false);
  if (oc != n_floats) {
    lb_gns_train_destroy(t);
    return lb_fail(LB_ERR_ARG, "segnn weight blob has %lld floats, expected %lld", (long long)n_floats, (long long)oc);
  }
  for (const lb_sgt_block& b : g->blocks)
    if (b.K > 256) {
      lb_gns_train_destroy(t);
      return lb_fail(LB_ERR_UNSUPPORTED, "segnn training: a tensor product with %d input channels (<= 256)", b.K);
    }
  t->n_floats = o;
  t->n_compact = oc;
  int rc = LB_OK;
  for (float** p : {&t->w, &t->g, &t->m, &t->v})
    if (!rc) rc = lb_alloc(p, (size_t)o);
  if (!rc) rc = lb_alloc(&t->loss_dev, 1);
  if (!rc) rc = lb_alloc(&t->dw_flag, (size_t)(1 + LB_DW_CALLS));
  if (!rc && hipMemset(t->dw_flag, 0, sizeof(int32_t) * (1 + LB_DW_CALLS)) != hipSuccess) rc = lb_fail(LB_ERR_HIP, "hipMemset");
  if (!rc) rc = lb_alloc(&t->cnt_dev, (size_t)e->g.B);
  if (!rc) {
    std::vector<float> padded((size_t)o, 0.f);
    for (int64_t i = 0; i < oc; ++i) padded[(size_t)t->cmap[(size_t)i]] = w[i];
    if (hipMemcpy(t->w, padded.data(), sizeof(float) * o, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemset(t->g, 0, sizeof(float) * o) != hipSuccess || hipMemset(t->m, 0, sizeof(float) * o) != hipSuccess ||
        hipMemset(t->v, 0, sizeof(float) * o) != hipSuccess)
      rc = lb_fail(LB_ERR_HIP, "weight upload failed");
  }
  if (rc) {
    lb_gns_train_destroy(t);
    return rc;
  }
  *out = t;
  return LB_OK;
}

// value_and_grad of _mse for SEGNN on the engine's CURRENT window / neighbor list: same contract as lb_gns_train_loss_grad
extern "C" int lb_segnn_train_loss_grad(lb_gns_train* t, const float* target_dev, float loss_weight, double* loss_out,
                                        float* pred_out_dev) {
  if (!t || !t->sg || !target_dev) return lb_fail(LB_ERR_ARG, "null argument / not a SEGNN training handle");
  return train_loss_grad_guarded(t, target_dev, loss_weight, loss_out, pred_out_dev);   // (X range guard: lb_train.hip)
}
static int segnn_train_loss_grad_once(lb_gns_train* t, const float* target_dev, float loss_weight, double* loss_out,
                                      float* pred_out_dev) {
  lb_engine* e = t->eng;
  lb_sgt* g = t->sg;
  if (e->e_cap <= 0) return lb_fail(LB_ERR_STATE, "lb_segnn_train_loss_grad before lb_nl_allocate");
  hipStream_t s = e->stream;
  LB_HIP(hipMemcpyAsync(e->ctrl_host, e->ctrl, sizeof(lb_ctrl), hipMemcpyDeviceToHost, s));
  LB_HIP(hipStreamSynchronize(s));
  if (e->ctrl_host->overflow_step >= 0) return lb_fail(LB_ERR_STATE, "neighbor list overflowed: re-allocate first");
  const int64_t E = e->ctrl_host->n_edges_total, BN = e->BN;
  const int B = g->desc.blocks_per_step, L = g->desc.num_mp_steps, C = g->desc.hidden, dim = e->g.dim;
  LB_TRY(sgt_ensure(t, BN, E));
  pack_all(t);
  t->red_tab.clear();
  t->red_off = 0;
  t->red_blocks = 0;
  LB_TRY(lbk_sg_prep(e, g->desc.homogeneous, g->desc.velocity_avg, g->node_ns4, g->node_nv4, g->xnode, g->eattr, g->msgsv,
                     g->nodesv, g->nattr, g->cap_e));
  // ---- forward (segnn.py:595-610)
  size_t bi = 0;
  {
    lb_sgt_op op = sgt_op(g->nodesv, g->node_ns, g->node_nv, nullptr, 0, nullptr, g->node_stride, g->node_ns4, g->node_nv4);
    LB_TRY(sgt_fwd(t, g->blocks[bi++], BN, &op, g->nattr, nullptr, g->f));
  }
  for (int k = 0; k < L; ++k) {
    const float* cur = nullptr;
    for (int i = 0; i < B; ++i) {
      float* dst = g->te[i & 1];
      if (i == 0) {
        lb_sgt_op ops[3] = {sgt_op(g->f, C, C, nullptr, 0, e->senders), sgt_op(g->f, C, C, nullptr, 0, e->receivers),
                            sgt_op(g->msgsv, 1, 1, nullptr, 0, nullptr, 16, 4, 4)};
        LB_TRY(sgt_fwd(t, g->blocks[bi++], E, ops, g->eattr, nullptr, dst));
      } else {
        lb_sgt_op op = sgt_op(cur, C, C);
        LB_TRY(sgt_fwd(t, g->blocks[bi++], E, &op, g->eattr, nullptr, dst));
      }
      cur = dst;
    }
    hipLaunchKernelGGL(k_seg_sum, GRID1(BN * 32), 0, s, e->row_ptr, cur ? cur : g->te[0], g->agg, BN, E, (const float*)nullptr,
                       (float*)nullptr);
    const float* ncur = nullptr;
    for (int i = 0; i < B; ++i) {
      const bool last = i == B - 1;
      float* dst = last ? g->f : g->tn[i & 1];
      if (i == 0) {
        lb_sgt_op ops[2] = {sgt_op(g->f, C, C), sgt_op(g->agg, C, C)};
        LB_TRY(sgt_fwd(t, g->blocks[bi++], BN, ops, g->nattr, last ? g->f : nullptr, dst));
      } else {
        lb_sgt_op op = sgt_op(ncur, C, C);
        LB_TRY(sgt_fwd(t, g->blocks[bi++], BN, &op, g->nattr, last ? g->f : nullptr, dst));
      }
      ncur = dst;
    }
  }
  {
    const float* ncur = g->f;
    for (int i = 0; i < B; ++i) {
      lb_sgt_op op = sgt_op(ncur, C, C);
      LB_TRY(sgt_fwd(t, g->blocks[bi++], BN, &op, g->nattr, nullptr, g->tn[i & 1]));
      ncur = g->tn[i & 1];
    }
    lb_sgt_op op = sgt_op(ncur, C, C);
    LB_TRY(sgt_fwd(t, g->blocks[bi++], BN, &op, g->nattr, nullptr, t->pred));
  }
  if (pred_out_dev) LB_HIP(hipMemcpyAsync(pred_out_dev, t->pred, sizeof(float) * BN * dim, hipMemcpyDeviceToDevice, s));
  LB_TRY(train_loss(t, t->pred, target_dev, loss_weight, t->dy));
  LB_TRY(train_sender_sort(t, E, BN));
  // ---- backward, blocks in reverse call order
  {
    lb_sgt_op op = sgt_op(nullptr, C, C, g->dtn[B & 1]);
    LB_TRY(sgt_bwd(t, g->blocks[--bi], BN, t->dy, nullptr, &op, g->nattr));            // output -> d readout_{B-1}
    for (int i = B - 1; i >= 0; --i) {
      float* dsrc = g->dtn[(i + 1) & 1];
      lb_sgt_op op2 = sgt_op(nullptr, C, C, i == 0 ? g->df : g->dtn[i & 1]);
      LB_TRY(sgt_bwd(t, g->blocks[--bi], BN, dsrc, nullptr, &op2, g->nattr));            // readout_i
    }
  }
  for (int k = L - 1; k >= 0; --k) {
    // update blocks: f_{k+1} = f_k + update(...): d update = df; df also flows to f_k through the residual (stays in df)
    const float* dsrc = g->df;
    for (int i = B - 1; i >= 0; --i) {
      if (i == 0) {
        lb_sgt_op ops[2] = {sgt_op(nullptr, C, C, g->df, 1), sgt_op(nullptr, C, C, g->dagg)};
        // (B = 1: dsrc IS df - k_sgt_out_bwd has consumed it into d raw before k_sgt_in_bwd accumulates into it)
        LB_TRY(sgt_bwd(t, g->blocks[--bi], BN, dsrc, nullptr, ops, g->nattr));
      } else {
        lb_sgt_op op = sgt_op(nullptr, C, C, g->dtn[i & 1]);
        LB_TRY(sgt_bwd(t, g->blocks[--bi], BN, dsrc, nullptr, &op, g->nattr));
        dsrc = g->dtn[i & 1];
      }
    }
    // message blocks: the last one's output feeds agg: d m[e] = dagg[rcv[e]]
    const float* esrc = g->dagg;
    const int32_t* eidx = e->receivers;
    for (int i = B - 1; i >= 0; --i) {
      if (i == 0) {
        lb_sgt_op ops[3] = {sgt_op(nullptr, C, C, g->dFs), sgt_op(nullptr, C, C, g->dFr), sgt_op(nullptr, 1, 1, nullptr)};
        LB_TRY(sgt_bwd(t, g->blocks[--bi], E, esrc, eidx, ops, g->eattr));
        if (E) hipLaunchKernelGGL(k_sgt_scatter, GRID1(BN * 32), 0, s, g->dFs, g->dFr, t->snd_ptr, t->snd_perm, e->row_ptr, g->df, BN, E);
      } else {
        lb_sgt_op op = sgt_op(nullptr, C, C, g->dte[i & 1]);
        LB_TRY(sgt_bwd(t, g->blocks[--bi], E, esrc, eidx, &op, g->eattr));
        esrc = g->dte[i & 1];
        eidx = nullptr;
      }
    }
  }
  {
    lb_sgt_op op = sgt_op(nullptr, g->node_ns, g->node_nv, nullptr);
    LB_TRY(sgt_bwd(t, g->blocks[--bi], BN, g->df, nullptr, &op, g->nattr));             // embedding: parameters only
  }
  if (bi != 0) return lb_fail(LB_ERR_STATE, "segnn training: block bookkeeping");
  LB_TRY(red_flush(t));
  LB_HIP(hipGetLastError());
  if (loss_out)   // (train_loss_grad_guarded synchronises and hands it over)
    LB_HIP(hipMemcpyAsync(status_loss(t), t->loss_dev, sizeof(double), hipMemcpyDeviceToHost, s));
  return LB_OK;
}
