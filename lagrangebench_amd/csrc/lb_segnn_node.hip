// lb_segnn_node.hip - the node-side ends of the SEGNN forward pass as ONE kernel each (round 4):
//   k_sg_embed:   SEGNN._transform's node part + O3Embedding     (lagrangebench/models/segnn.py:536-575, 184-215)
//                 = round 3's k_sg_edge_prep + k_sg_node_prep + k_sg_tp<PLAIN>
//   k_sg_readout: O3Decoder (two gated blocks + the 1x1o output block) + the rollout step's integrator
//                 (segnn.py:218-249; case.py:230-259) = round 3's 2 x k_sg_tp<GATE> + k_sg_tp<OUTVEC> + k_integrate
// On one DAM2D trajectory those seven launches were 60 us of a 0.42 ms step.  Both kernels work on tiles of 16
// consecutive nodes in the register-chained f16x2 scheme of lb_segnn_dev.h (lane (n = l&15, g = l>>4) holds, for node
// n, the channels 16 mb + 4 g + j of an SV row [s | vx | vy | vz]).
//
// k_sg_embed forms the operand row in registers: the node's SV row is a permutation of the engine's fp32 feature row
// ([vel_hist | vel_mag | bound | force], written by the neighbor-search waves) plus the particle-type one-hot, so a
// lane fetches its 8 (+ 8 per component) entries straight from that row; the node attribute = SH(velocity) + mean over
// the incoming edges of SH(rel_disp) is summed by the four lanes of a node over the edge features the search wrote
// (no eattr / msgsv arrays on this path).
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "lb_segnn_dev.h"
#include "lb_features.h"

// ---------------------------------------------------------------------------------------------- embedding
#define SGE_WS 0      // K=64 ([32 scalar slots | 32 vector slots]) x M=32: 2 x 2 x 2 x 64
#define SGE_WT 512    // K=32 x M=32
#define SGE_WV 768
#define SGE_VEC 1024  // bias (8 f32x4)
#define SGE_IMAGE 1032

struct lb_sg_embed_args {
  const lb_ctrl* ctrl;
  int64_t n_rows;
  const float* xnode;      // [rows][32] engine feature row
  const int32_t* ptype;
  const int32_t* row_ptr;
  const float* efeat;      // [E][8]
  const float* image;
  float* f;                // [rows][128] hidden state (out)
  float* nattr;            // [rows][4] (out)
  int32_t K, dim, has_vel_mag, has_bound, has_force, homogeneous, vel_avg;
};

template <int DIM>
__global__ void __launch_bounds__(512, 2) k_sg_embed(lb_sg_embed_args a) {
  constexpr int NC = DIM, NT = 512, WAVES = 8;
  __shared__ f32x4 sW[SGE_IMAGE];
  const int poisoned = a.ctrl->overflow_step;
  const int E = a.ctrl->n_edges_total;
  const int tid = threadIdx.x;
  constexpr int NST = (SGE_IMAGE + NT - 1) / NT;
  f32x4 st[NST];
  {
    const f32x4* src = reinterpret_cast<const f32x4*>(a.image);
#pragma unroll
    for (int k = 0; k < NST; ++k) {
      const int i = tid + k * NT;
      st[k] = src[i < SGE_IMAGE ? i : SGE_IMAGE - 1];
    }
  }
  if (poisoned >= 0) return;
#pragma unroll
  for (int k = 0; k < NST; ++k) {
    const int i = tid + k * NT;
    if (i < SGE_IMAGE) sW[i] = st[k];
  }
  __syncthreads();
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, g = lane >> 4;
  const int ntiles = (int)((a.n_rows + 15) >> 4);
  const lds_cptr w0 = (lds_cptr)(sW + lane), vec = (lds_cptr)(sW + SGE_VEC + g);
  const int K = a.K, dim = a.dim;
  const int col_mag = K * dim, col_bnd = col_mag + (a.has_vel_mag ? K : 0), col_frc = col_bnd + (a.has_bound ? 2 * dim : 0);
  const int ns_mag = a.has_vel_mag ? K : 0;
  for (int t = blockIdx.x * WAVES + wave; t < ntiles; t += gridDim.x * WAVES) {
    const int64_t row = (int64_t)t * 16 + n;
    const bool valid = row < a.n_rows;
    const int64_t rl = valid ? row : a.n_rows - 1;
    const float* x = a.xnode + rl * 32;
    int k0 = a.row_ptr[rl], k1 = a.row_ptr[rl + 1];
    int pt = a.homogeneous ? 0 : a.ptype[rl];
    if (pt < 0) pt += 9;
    // ---- the operand row, slot by slot (k_sg_node_prep's mapping)
    auto scalar_slot = [&](int s) -> float {  // s in [0, 32)
      if (s < ns_mag) return x[col_mag + s];
      if (!a.homogeneous && s < ns_mag + 9) return (s - ns_mag) == pt ? 1.f : 0.f;
      return 0.f;
    };
    auto vector_slot = [&](int v, int d) -> float {  // channel v in [0, 32), component d < dim
      if (v < K) return x[v * dim + d];
      int w = v - K;
      if (a.has_bound) {
        if (w < 2) return x[col_bnd + w * dim + d];
        w -= 2;
      }
      if (a.has_force && w == 0) return x[col_frc + d];
      return 0.f;
    };
    f32x4 X[8];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        X[h][j] = scalar_slot(16 * h + 4 * g + j);
#pragma unroll
        for (int c = 0; c < NC; ++c) X[2 + 2 * c + h][j] = vector_slot(16 * h + 4 * g + j, c);
      }
    }
    // ---- node attribute: SH(velocity) + mean over the incoming edges of SH(rel_disp); l = 0 entry forced to 1
    float vm[3] = {0.f, 0.f, 0.f};
    for (int tt = a.vel_avg ? 0 : K - 1; tt < K; ++tt)
#pragma unroll
      for (int d = 0; d < NC; ++d) vm[d] += x[tt * dim + d];
    if (a.vel_avg && K > 1)
#pragma unroll
      for (int d = 0; d < 3; ++d) vm[d] = vm[d] / (float)K;
    const float vn = sqrtf(vm[0] * vm[0] + vm[1] * vm[1] + vm[2] * vm[2]);
    const float vinv = vn == 0.f ? 0.f : SG_Y1 / vn;
    k0 = k0 < E ? k0 : E;
    k1 = k1 < E ? k1 : E;
    float acc[3] = {0.f, 0.f, 0.f};
    for (int k = k0 + g; k < k1; k += 4) {
      const f32x4 ef = reinterpret_cast<const f32x4*>(a.efeat)[2 * (int64_t)k];
      const float dx = ef[0], dy = ef[1], dz = DIM == 3 ? ef[2] : 0.f;
      const float nrm = sqrtf(dx * dx + dy * dy + dz * dz);
      const float inv = nrm == 0.f ? 0.f : SG_Y1 / nrm;
      acc[0] += dx * inv;
      acc[1] += dy * inv;
      acc[2] += dz * inv;
    }
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      acc[d] += __shfl_xor(acc[d], 16);
      acc[d] += __shfl_xor(acc[d], 32);
    }
    const float cnt = (float)((k1 - k0) > 1 ? (k1 - k0) : 1);
    const float at[3] = {vm[0] * vinv + acc[0] / cnt, vm[1] * vinv + acc[1] / cnt, vm[2] * vinv + acc[2] / cnt};
    if (valid && g == 0) reinterpret_cast<f32x4*>(a.nattr)[row] = f32x4{1.f, at[0], at[1], at[2]};
    // ---- O3TensorProduct (no gate): f = [s | (v.a)/sqrt3] Ws + b, [s a_c | v_c] Wv
    f32x4 S[4], T[2], V[3][2];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      S[m] = vec[4 * m];
      T[m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < 3; ++c) V[c][m] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    sg_operand<NC, 2, 2, false>(w0 + SGE_WS, w0 + SGE_WT, w0 + SGE_WV, X, at, S, T, V);
    if (valid) {
      f32x4* frow = reinterpret_cast<f32x4*>(a.f) + row * 32 + g;
      frow[0] = S[0];
      frow[4] = S[1];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        if (c < NC) {
          frow[4 * (2 + 2 * c)] = V[c][0] + T[0] * at[c];
          frow[4 * (3 + 2 * c)] = V[c][1] + T[1] * at[c];
        } else {  // 2D: the z component of the hidden state is exactly zero, and stays so (k_sg_upd never touches it)
          frow[4 * (2 + 2 * c)] = f32x4{0.f, 0.f, 0.f, 0.f};
          frow[4 * (3 + 2 * c)] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
      }
    }
  }
}

// ws (ktrue x 32), wv (ktrue x 32), b (32): rows [ns scalars | nv vectors] (oracle order), ktrue = ns + nv
void lb_sg_embed_image(const float* ws, const float* wv, const float* b, int ns, int nv, float* out) {
  const float sc = 1.0f / sqrtf((float)(ns + nv)), is3 = 0.5773502691896258f;
  memset(out, 0, sizeof(float) * SGE_IMAGE * 4);
  std::vector<float> m;
  auto pack = [&](int K, int M, int off) { lb_pack_weight16h(m.data(), K, M, K, out + (size_t)off * 4, M); };
  m.assign(64 * 32, 0.f);
  for (int k = 0; k < ns; ++k)
    for (int j = 0; j < 32; ++j) m[k * 32 + j] = ws[k * 32 + j] * sc;  // node attribute a0 == 1
  for (int k = 0; k < nv; ++k)
    for (int j = 0; j < 32; ++j) m[(32 + k) * 32 + j] = ws[(ns + k) * 32 + j] * (is3 * sc);
  pack(64, 32, SGE_WS);
  m.assign(32 * 32, 0.f);
  for (int k = 0; k < ns; ++k)
    for (int j = 0; j < 32; ++j) m[k * 32 + j] = wv[k * 32 + j] * sc;
  pack(32, 32, SGE_WT);
  m.assign(32 * 32, 0.f);
  for (int k = 0; k < nv; ++k)
    for (int j = 0; j < 32; ++j) m[k * 32 + j] = wv[(ns + k) * 32 + j] * sc;
  pack(32, 32, SGE_WV);
  float* v = out + (size_t)SGE_VEC * 4;
  for (int j = 0; j < 32; ++j) v[j] = b[j];
}
int lb_sg_embed_image_floats(void) { return SGE_IMAGE * 4; }

int lbk_sg_embed(lb_engine* e, const float* xnode, const float* image, float* f, float* nattr, int homogeneous,
                 int vel_avg) {
  lb_sg_embed_args a{};
  a.ctrl = e->ctrl;
  a.n_rows = e->BN;
  a.xnode = xnode;
  a.ptype = e->ptype;
  a.row_ptr = e->row_ptr;
  a.efeat = e->efeat;
  a.image = image;
  a.f = f;
  a.nattr = nattr;
  a.K = e->g.isl - 1;
  a.dim = e->g.dim;
  a.has_vel_mag = e->g.has_vel_mag;
  a.has_bound = e->g.has_bound;
  a.has_force = e->g.force_kind != LB_FORCE_NONE;
  a.homogeneous = homogeneous;
  a.vel_avg = vel_avg;
  const int ntiles = (int)((e->BN + 15) / 16);
  const int nb = std::min(512, (ntiles + 7) / 8);
  if (e->g.dim == 2)
    hipLaunchKernelGGL((k_sg_embed<2>), dim3(nb), dim3(512), 0, e->stream, a);
  else
    hipLaunchKernelGGL((k_sg_embed<3>), dim3(nb), dim3(512), 0, e->stream, a);
  LB_HIP(hipGetLastError());
  return LB_OK;
}

// ------------------------------------------------------------------------------------------------ readout
#define SGR_WS0 0     // K=64 x M=64: 2 x 4 x 2 x 64
#define SGR_WT0 1024  // K=32 x M=32
#define SGR_WV0 1280
#define SGR_WS1 1536
#define SGR_WT1 2560
#define SGR_WV1 2816
#define SGR_VEC 3072  // b0 (16 f32x4), b1 (16), wo_s (8), wo_v (8)
#define SGR_IMAGE 3120

struct lb_sg_readout_args {
  const lb_ctrl* ctrl;
  int64_t n_rows;
  const float* f;       // [rows][128]
  const float* nattr;   // [rows][4]
  const float* image;
  float* acc_out;       // [rows][4]
  lb_integ_job integ;   // rollout step: integrate_fn + window advance in the epilogue (on = 0: stand-alone forward)
};

template <int DIM>
__global__ void __launch_bounds__(512, 2) k_sg_readout(lb_geom geom, lb_sg_readout_args a) {
  constexpr int NC = DIM, NT = 512, WAVES = 8, NMB = 2 + 2 * NC;
  __shared__ f32x4 sW[SGR_IMAGE];
  const int poisoned = a.ctrl->overflow_step;
  const int step = a.ctrl->step;
  const int tid = threadIdx.x;
  constexpr int NST = (SGR_IMAGE + NT - 1) / NT;
  f32x4 st[NST];
  {
    const f32x4* src = reinterpret_cast<const f32x4*>(a.image);
#pragma unroll
    for (int k = 0; k < NST; ++k) {
      const int i = tid + k * NT;
      st[k] = src[i < SGR_IMAGE ? i : SGR_IMAGE - 1];
    }
  }
  if (poisoned >= 0) return;
#pragma unroll
  for (int k = 0; k < NST; ++k) {
    const int i = tid + k * NT;
    if (i < SGR_IMAGE) sW[i] = st[k];
  }
  __syncthreads();
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, g = lane >> 4;
  const int ntiles = (int)((a.n_rows + 15) >> 4);
  const lds_cptr w0 = (lds_cptr)(sW + lane), vec = (lds_cptr)(sW + SGR_VEC + g);
  for (int t = blockIdx.x * WAVES + wave; t < ntiles; t += gridDim.x * WAVES) {
    const int64_t row = (int64_t)t * 16 + n;
    const bool valid = row < a.n_rows;
    const int64_t rl = valid ? row : a.n_rows - 1;
    const f32x4* frow = reinterpret_cast<const f32x4*>(a.f) + rl * 32 + g;
    f32x4 X[8];
#pragma unroll
    for (int mb = 0; mb < NMB; ++mb) X[mb] = frow[4 * mb];
    const f32x4 na = reinterpret_cast<const f32x4*>(a.nattr)[rl];
    const float at[3] = {na[1], na[2], na[3]};
    lb_integ_in pre{};
    if (a.integ.on && g == 0) pre = lb_integrate_fetch(geom, a.n_rows, a.integ.win, step, a.integ.ptype, rl);
    f32x4 S[4], T[2], V[3][2], H[8];
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
#pragma unroll
      for (int m = 0; m < 4; ++m) S[m] = vec[16 * blk + 4 * m];
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        T[m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 3; ++c) V[c][m] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      if (blk == 0) {
        sg_operand<NC, 4, 4, false>(w0 + SGR_WS0, w0 + SGR_WT0, w0 + SGR_WV0, X, at, S, T, V);
        sg_gate<NC>(S, T, V, at, H);
      } else {
        sg_operand<NC, 4, 4, false>(w0 + SGR_WS1, w0 + SGR_WT1, w0 + SGR_WV1, H, at, S, T, V);
        sg_gate<NC>(S, T, V, at, X);
      }
    }
    // ---- output block 32x0e+32x1o -> 1x1o: out_c = a_c (s . wo_s) + (v_c . wo_v); fp32, 4 lanes per node
    float ps = 0.f, pv[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const f32x4 ws = vec[32 + 4 * h], wv = vec[40 + 4 * h];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        ps += X[h][j] * ws[j];
#pragma unroll
        for (int c = 0; c < NC; ++c) pv[c] += X[2 + 2 * c + h][j] * wv[j];
      }
    }
    ps += __shfl_xor(ps, 16);
    ps += __shfl_xor(ps, 32);
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      pv[c] += __shfl_xor(pv[c], 16);
      pv[c] += __shfl_xor(pv[c], 32);
    }
    if (valid && g == 0) {
      const f32x4 o = {at[0] * ps + pv[0], at[1] * ps + pv[1], NC == 3 ? at[2] * ps + pv[2] : 0.f, 0.f};
      reinterpret_cast<f32x4*>(a.acc_out)[row] = o;
      if (a.integ.on) {
        const float av[4] = {o[0], o[1], o[2], 0.f};
        lb_integrate_body(geom, a.n_rows, a.integ.win, step, a.integ.ptype, av, nullptr, a.integ.traj, a.integ.T,
                          a.integ.pred, a.integ.pred_T, row, &pre);
      }
    }
  }
  // the step counter is advanced by the LAST workgroup to finish (k_integrate's job)
  if (a.integ.on) {
    __syncthreads();
    if (tid == 0 && atomicAdd(a.integ.blocks_done, 1) == (int)gridDim.x - 1) {
      *a.integ.blocks_done = 0;
      const_cast<lb_ctrl*>(a.ctrl)->step = step + 1;
    }
  }
}

// readout block i: ws_i (64 x 64), wv_i (64 x 32), b_i (64); output block: wo (64 x 1): rows [32 scalars | 32 vectors]
void lb_sg_readout_image(const float* ws0, const float* wv0, const float* b0, const float* ws1, const float* wv1,
                         const float* b1, const float* wo, float* out) {
  const float sc = 1.0f / sqrtf(64.f), is3 = 0.5773502691896258f;
  const float zs = SG_NL2E, cg = SG_C_SIGMOID, ks = SG_K_SILU;
  memset(out, 0, sizeof(float) * SGR_IMAGE * 4);
  std::vector<float> m;
  auto pack = [&](int K, int M, int off) { lb_pack_weight16h(m.data(), K, M, K, out + (size_t)off * 4, M); };
  for (int blk = 0; blk < 2; ++blk) {
    const float* ws = blk ? ws1 : ws0;
    const float* wv = blk ? wv1 : wv0;
    const float in_s = blk ? ks : 1.f;  // block 1 consumes block 0's scalars as z sigma(z)
    m.assign(64 * 64, 0.f);
    for (int k = 0; k < 64; ++k) {
      const float f = ((k >= 32) ? is3 * sc : sc * in_s) * zs;  // node attribute a0 == 1
      for (int j = 0; j < 64; ++j) m[k * 64 + j] = ws[k * 64 + j] * f;
    }
    pack(64, 64, blk ? SGR_WS1 : SGR_WS0);
    m.assign(32 * 32, 0.f);
    for (int k = 0; k < 32; ++k)
      for (int j = 0; j < 32; ++j) m[k * 32 + j] = wv[k * 32 + j] * (sc * in_s * cg);
    pack(32, 32, blk ? SGR_WT1 : SGR_WT0);
    for (int k = 0; k < 32; ++k)
      for (int j = 0; j < 32; ++j) m[k * 32 + j] = wv[(32 + k) * 32 + j] * (sc * cg);
    pack(32, 32, blk ? SGR_WV1 : SGR_WV0);
  }
  float* v = out + (size_t)SGR_VEC * 4;
  for (int j = 0; j < 64; ++j) {
    v[j] = b0[j] * zs;
    v[64 + j] = b1[j] * zs;
  }
  for (int j = 0; j < 32; ++j) {
    v[128 + j] = wo[j] * (sc * ks);   // scalars of the second gated block arrive as z sigma(z)
    v[160 + j] = wo[32 + j] * sc;
  }
}
int lb_sg_readout_image_floats(void) { return SGR_IMAGE * 4; }

int lbk_sg_readout(lb_engine* e, const float* f, const float* nattr, const float* image, float* acc_out) {
  lb_sg_readout_args a{};
  a.ctrl = e->ctrl;
  a.n_rows = e->BN;
  a.f = f;
  a.nattr = nattr;
  a.image = image;
  a.acc_out = acc_out;
  if (e->integ_job.on) {  // rollout step: the integrator rides along
    a.integ = e->integ_job;
    e->integ_done = true;
  }
  const int ntiles = (int)((e->BN + 15) / 16);
  const int nb = std::min(512, (ntiles + 7) / 8);
  if (e->g.dim == 2)
    LB_LAUNCH_TIMED(e, (k_sg_readout<2>), dim3(nb), dim3(512), e->g, a);
  else
    LB_LAUNCH_TIMED(e, (k_sg_readout<3>), dim3(nb), dim3(512), e->g, a);
  LB_HIP(hipGetLastError());
  return LB_OK;
}
