// lb_gns_generic.hip - GNS with num_mlp_layers != 2 (models/utils.py:100-115 build_mlp: hk.nets.MLP of
// [latent]*(n-1) + [out] Linears, ReLU between them, LayerNorm after the last; models/gns.py:65-133).
//
// The fused kernels (lb_edge16v / lb_node16s / lb_edge16 / lb_node16h) hard-wire the published shape
// "two Linears per MLP": both weight matrices of an MLP live in LDS and the hidden activations never
// leave the registers.  Any other depth runs here, on ONE kernel that applies a single 128x128 Linear
// to 16-row tiles,
//     y = [LayerNorm]( W^T relu?(x) + bias + add0[idx0[row]] + add1[idx1[row]] ) [+ residual],
// composed by lbk_gns_forward_generic into the same encode - process - decode graph:
//   * a first Linear over a concatenated input is split by input block, exactly like the fused path:
//     edge MLP  [n_s | n_r | e] W0 = (n Ws)[senders] + (n Wr)[receivers] + e We   (two node-sized
//     Linears, then the edge Linear gathers their rows as addends);  node MLP  [n | agg] W0 likewise;
//   * hidden activations are stored BEFORE the ReLU; the consumer applies it while splitting its
//     operand (lb_gemm16v<RELU>), so a chain of n Linears is n launches reading/writing 128 floats per
//     row - about 2-3x the HBM traffic of the fused two-Linear kernels, the price of generality;
//   * jraph.segment_sum is the stand-alone k_segment_sum on the receiver-sorted list.
// Arithmetic: the same f16x2 split products (or exact fp32 MFMA under LB_MATH=f32 / after the range
// guard fired) and the same LayerNorm code as the fused kernels, so the 1e-5 parity bar applies.
// A latent narrower than 128 is zero-padded exactly as in lb_gns_create (lb_ctrl::ln_inv_d / ln_pad).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "lb_f16x2.h"
#include "lb_features.h"

#define MFMA16F(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
#define GD_THREADS 512
#define GD_WAVES 8

struct lb_dense_args {
  const lb_ctrl* ctrl;
  int64_t n_rows;        // < 0: the device-side edge count ctrl->n_edges_total
  const float* x;        // input rows
  int x_blocked;         // 1: tile-blocked edge-latent layout, 0: row-major with x_stride floats per row
  int x_stride, x_width; // row-major only: columns >= x_width read as 0 (narrow encoder inputs)
  const float* w;        // packed 128 x 128 (f16x2 hi|lo fragments, or fp32 fragments)
  const float* bias;     // [128] or null
  const float* add0;     // optional addend rows (row-major, 128 per row), gathered through idx0 (null = row)
  const int32_t* idx0;
  const float* add1;
  const int32_t* idx1;
  const float* ln_s;     // LayerNorm scale / offset, null = no LayerNorm
  const float* ln_o;
  const float* resid;    // optional residual added AFTER the LayerNorm
  int resid_blocked;
  float* out;            // 128 per row
  int out_blocked;
  float* out_pre;        // optional row-major copy of the result BEFORE the residual (the message)
  float* acc_out;        // decoder head: first 4 columns as one f32x4 per row (out may be null)
  int out_dim;
};

// acc[0..7] += W^T v over 8 k-blocks of 16 on the fp32 MFMA (weights packed by lb_pack_weight16)
__device__ __forceinline__ void lb_gemm16f(lds_cptr wl, const f32x4 (&v)[8], f32x4 (&acc)[8]) {
  f32x4 a0 = wl[0], a1 = wl[64];
#pragma unroll
  for (int step = 0; step < 32; ++step) {
    f32x4 n0 = a0, n1 = a1;
    if (step + 1 < 32) {
      n0 = wl[((step + 1) * 2) * 64];
      n1 = wl[((step + 1) * 2 + 1) * 64];
    }
    const float b = v[step >> 2][step & 3];
    acc[0] = MFMA16F(a0[0], b, acc[0]);
    acc[1] = MFMA16F(a0[1], b, acc[1]);
    acc[2] = MFMA16F(a0[2], b, acc[2]);
    acc[3] = MFMA16F(a0[3], b, acc[3]);
    acc[4] = MFMA16F(a1[0], b, acc[4]);
    acc[5] = MFMA16F(a1[1], b, acc[5]);
    acc[6] = MFMA16F(a1[2], b, acc[6]);
    acc[7] = MFMA16F(a1[3], b, acc[7]);
    a0 = n0;
    a1 = n1;
    SB();
  }
}

template <bool F16, bool RELU_IN>
__global__ void __launch_bounds__(GD_THREADS, 2) k_dense16(lb_dense_args a) {
  __shared__ f32x4 sW[4096 + 96];  // packed weights | bias | ln scale | ln offset
  const int poisoned = a.ctrl->overflow_step;  // acted on after the staging loads are in flight
  const int n_edges = a.ctrl->n_edges_total;
  const int tid = threadIdx.x;
  {
    const f32x4* gw = reinterpret_cast<const f32x4*>(a.w);
    for (int i = tid; i < 4096; i += GD_THREADS) sW[i] = gw[i];
    if (tid < 96) {
      const float* src = tid < 32 ? a.bias : (tid < 64 ? a.ln_s : a.ln_o);
      sW[4096 + tid] = src ? reinterpret_cast<const f32x4*>(src)[tid & 31] : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  if (poisoned >= 0) return;
  __syncthreads();
  const int64_t R = a.n_rows < 0 ? (int64_t)n_edges : a.n_rows;
  const int ntiles = (int)((R + 15) >> 4);
  const int lane = tid & 63, wave = tid >> 6;
  const int n = lane & 15, g = lane >> 4;
  const lds_cptr lw = (lds_cptr)sW + lane;
  const lds_cptr lb = (lds_cptr)sW + 4096 + g;
  const lds_cptr lns = (lds_cptr)sW + 4096 + 32 + g;
  const lds_cptr lno = (lds_cptr)sW + 4096 + 64 + g;
  const float inv_d = a.ctrl->ln_inv_d, pad = a.ctrl->ln_pad;
  bool probed = false;
  for (int t = blockIdx.x * GD_WAVES + wave; t < ntiles; t += gridDim.x * GD_WAVES) {
    const int64_t row = (int64_t)t * 16 + n;
    const bool valid = row < R;
    const int64_t rowc = valid ? row : R - 1;
    f32x4 v[8], acc[8];
    if (a.x_blocked) {
      const f32x4* xb = reinterpret_cast<const f32x4*>(a.x) + (int64_t)t * 512 + lane;
#pragma unroll
      for (int mb = 0; mb < 8; ++mb) v[mb] = valid ? xb[64 * mb] : f32x4{0.f, 0.f, 0.f, 0.f};
    } else {
      const float* xr = a.x + rowc * a.x_stride + 4 * g;
#pragma unroll
      for (int mb = 0; mb < 8; ++mb)
        v[mb] = (16 * mb + 4 * g < a.x_width) ? *reinterpret_cast<const f32x4*>(xr + 16 * mb)
                                              : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int mb = 0; mb < 8; ++mb) acc[mb] = lb[4 * mb];
    if (a.add0) {
      const int64_t r0 = a.idx0 ? (int64_t)a.idx0[rowc] : rowc;
      const f32x4* p = reinterpret_cast<const f32x4*>(a.add0) + r0 * 32 + g;
#pragma unroll
      for (int mb = 0; mb < 8; ++mb) acc[mb] = acc[mb] + p[4 * mb];
    }
    if (a.add1) {
      const int64_t r1 = a.idx1 ? (int64_t)a.idx1[rowc] : rowc;
      const f32x4* p = reinterpret_cast<const f32x4*>(a.add1) + r1 * 32 + g;
#pragma unroll
      for (int mb = 0; mb < 8; ++mb) acc[mb] = acc[mb] + p[4 * mb];
    }
    if (F16) {
      if (!probed) {
        lb_range_probe(a.ctrl, v, 8);
        probed = true;
      }
      lb_gemm16v<RELU_IN, 4>(lw, v, acc);
    } else {
      if (RELU_IN) {
#pragma unroll
        for (int mb = 0; mb < 8; ++mb)
#pragma unroll
          for (int j = 0; j < 4; ++j) v[mb][j] = fmaxf(v[mb][j], 0.f);
      }
      lb_gemm16f(lw, v, acc);
    }
    f32x4 y[8];
    if (a.ln_s)
      lb_layernorm16<true>(acc, lns, lno, y, inv_d, pad);
    else
      lb_layernorm16<false>(acc, lns, lno, y);
    if (a.out_pre && valid) {
      f32x4* pr = reinterpret_cast<f32x4*>(a.out_pre) + row * 32 + g;
#pragma unroll
      for (int mb = 0; mb < 8; ++mb) pr[4 * mb] = y[mb];
    }
    if (a.resid) {
      if (a.resid_blocked) {
        const f32x4* rb = reinterpret_cast<const f32x4*>(a.resid) + (int64_t)t * 512 + lane;
#pragma unroll
        for (int mb = 0; mb < 8; ++mb) y[mb] = y[mb] + rb[64 * mb];
      } else {
        const f32x4* rr = reinterpret_cast<const f32x4*>(a.resid) + rowc * 32 + g;
#pragma unroll
        for (int mb = 0; mb < 8; ++mb) y[mb] = y[mb] + rr[4 * mb];
      }
    }
    if (a.out) {
      if (a.out_blocked) {
        f32x4* ob = reinterpret_cast<f32x4*>(a.out) + (int64_t)t * 512 + lane;
#pragma unroll
        for (int mb = 0; mb < 8; ++mb) ob[64 * mb] = y[mb];
      } else if (valid) {
        f32x4* orow = reinterpret_cast<f32x4*>(a.out) + row * 32 + g;
#pragma unroll
        for (int mb = 0; mb < 8; ++mb) orow[4 * mb] = y[mb];
      }
    }
    if (a.acc_out && valid && g == 0) {
      const f32x4 o = y[0];
      reinterpret_cast<f32x4*>(a.acc_out)[row] = o;
      bool bad = false;
      for (int d = 0; d < a.out_dim; ++d) bad |= !(fabsf(o[d]) <= 3.0e38f);
      if (bad) lb_raise_math(a.ctrl, LB_MATH_NONFINITE);
    }
  }
}

static int lbk_dense16(lb_engine* e, const lb_dense_args& a, int64_t rows_bound, bool relu_in) {
  const int64_t tiles = std::max<int64_t>(1, (rows_bound + 15) / 16);
  const int grid = (int)std::min<int64_t>(512, (tiles + GD_WAVES - 1) / GD_WAVES);
  if (e->f16x2) {
    if (relu_in)
      hipLaunchKernelGGL((k_dense16<true, true>), dim3(grid), dim3(GD_THREADS), 0, e->stream, a);
    else
      hipLaunchKernelGGL((k_dense16<true, false>), dim3(grid), dim3(GD_THREADS), 0, e->stream, a);
  } else {
    if (relu_in)
      hipLaunchKernelGGL((k_dense16<false, true>), dim3(grid), dim3(GD_THREADS), 0, e->stream, a);
    else
      hipLaunchKernelGGL((k_dense16<false, false>), dim3(grid), dim3(GD_THREADS), 0, e->stream, a);
  }
  LB_HIP(hipGetLastError());
  return LB_OK;
}

// ---------------------------------------------------------------------------------- decoder
// k_decoder16: the decoder MLP of the default depth (gns.py:125-133: Linear 128 -> 128, ReLU, Linear 128 ->
// out_dim, no LayerNorm) on the 16-row tile scheme: W0 and the one 16-column block of W1 that holds the
// out_dim outputs resident in LDS, node latents read once (row-major), the hidden layer stays in registers,
// one f32x4 per node written.  Replaces round 1's one-wave-per-32-rows fp32 kernel (k_decoder: 45 us per
// launch on 64 k nodes, weights re-read from global memory by every wave).
struct lb_dec16_args {
  const lb_ctrl* ctrl;
  int64_t n_rows;
  const float* nlat;
  const float* w0;   // packed 128 x 128 (f16x2 hi|lo or fp32 fragments)
  const float* b0;   // [128]
  const float* w1;   // f16x2: packed 128 x 16 (lb_pack_weight16h, Mpad 16); fp32: packed 128 x 128, block 0 used
  const float* b1;   // [>= 4]
  float unscale;     // result of the w1 product is multiplied by this (power of two) before b1 is added
  float* acc_out;    // [rows][4]
  int out_dim;
  lb_integ_job integ;  // rollout step: integrate_fn + window advance in the epilogue (on = 0: stand-alone forward)
};

template <bool F16>
__global__ void __launch_bounds__(GD_THREADS, 2) k_decoder16(lb_dec16_args a, lb_geom geom) {
  constexpr int NW1 = F16 ? 512 : 4096;
  __shared__ f32x4 sW[4096 + NW1 + 33];
  const int poisoned = a.ctrl->overflow_step;  // acted on after the staging loads are in flight
  const int step = a.ctrl->step;
  const int tid = threadIdx.x;
  {
    const f32x4* g0 = reinterpret_cast<const f32x4*>(a.w0);
    const f32x4* g1 = reinterpret_cast<const f32x4*>(a.w1);
    for (int i = tid; i < 4096; i += GD_THREADS) sW[i] = g0[i];
    for (int i = tid; i < NW1; i += GD_THREADS) sW[4096 + i] = g1[i];
    if (tid < 32) sW[4096 + NW1 + tid] = reinterpret_cast<const f32x4*>(a.b0)[tid];
    if (tid == 32) sW[4096 + NW1 + 32] = reinterpret_cast<const f32x4*>(a.b1)[0];
  }
  if (poisoned >= 0) return;
  __syncthreads();
  const int ntiles = (int)((a.n_rows + 15) >> 4);
  const int lane = tid & 63, wave = tid >> 6;
  const int n = lane & 15, g = lane >> 4;
  const lds_cptr lw0 = (lds_cptr)sW + lane, lw1 = (lds_cptr)sW + 4096 + lane;
  const lds_cptr lb0 = (lds_cptr)sW + 4096 + NW1 + g;
  const f32x4 b1v = sW[4096 + NW1 + 32];
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  bool probed = false;
  for (int t = blockIdx.x * GD_WAVES + wave; t < ntiles; t += gridDim.x * GD_WAVES) {
    const int64_t row = (int64_t)t * 16 + n;
    const bool valid = row < a.n_rows;
    const int64_t rowc = valid ? row : a.n_rows - 1;
    f32x4 v[8], acc[8];
    // rollout step: what the integrator reads of this row goes out with the tile's other loads
    lb_integ_in pre{};
    if (a.integ.on && valid && g == 0) pre = lb_integrate_fetch(geom, a.n_rows, a.integ.win, step, a.integ.ptype, row);
    const f32x4* xr = reinterpret_cast<const f32x4*>(a.nlat) + rowc * 32 + g;
#pragma unroll
    for (int mb = 0; mb < 8; ++mb) v[mb] = xr[4 * mb];
#pragma unroll
    for (int mb = 0; mb < 8; ++mb) acc[mb] = lb0[4 * mb];
    f32x4 o = zero;  // C layout: lane (n, g) holds outputs 4g .. 4g+3 of row n
    if constexpr (F16) {
      if (!probed) {
        lb_range_probe(a.ctrl, v, 8);
        probed = true;
      }
      lb_gemm16v<false, 4>(lw0, v, acc);
      if (t < (int)gridDim.x * GD_WAVES) lb_range_probe(a.ctrl, acc, 8);
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        f32x4 r0, r1;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          r0[j] = fmaxf(acc[2 * p][j], 0.f);
          r1[j] = fmaxf(acc[2 * p + 1][j], 0.f);
        }
        h8 bh, bl;
        lb_split8v(r0, r1, bh, bl);
        const h8 ah = __builtin_bit_cast(h8, lw1[(p * 2 + 0) * 64]);
        const h8 al = __builtin_bit_cast(h8, lw1[(p * 2 + 1) * 64]);
        o = MFMA16H(al, bh, o);
        o = MFMA16H(ah, bl, o);
        o = MFMA16H(ah, bh, o);
      }
    } else {
      lb_gemm16f(lw0, v, acc);
#pragma unroll
      for (int step = 0; step < 32; ++step) {
        const f32x4 a0 = lw1[(step * 2) * 64];
        o = MFMA16F(a0[0], fmaxf(acc[step >> 2][step & 3], 0.f), o);
      }
    }
    if (valid && g == 0) {
      o = o * a.unscale + b1v;
      reinterpret_cast<f32x4*>(a.acc_out)[row] = o;
      bool bad = false;
      for (int d = 0; d < a.out_dim; ++d) bad |= !(fabsf(o[d]) <= 3.0e38f);
      if (bad) lb_raise_math(a.ctrl, LB_MATH_NONFINITE);
      if (a.integ.on) {
        const float av[4] = {o[0], o[1], o[2], o[3]};
        lb_integrate_body(geom, a.n_rows, a.integ.win, step, a.integ.ptype, av, nullptr, a.integ.traj, a.integ.T,
                          a.integ.pred, a.integ.pred_T, row, &pre);
      }
    }
  }
  // the step counter is advanced by the LAST workgroup to finish (k_integrate's job; every workgroup has made its last
  // read of ctrl->step - the range guard's - by then)
  if (a.integ.on) {
    __syncthreads();
    if (tid == 0 && atomicAdd(a.integ.blocks_done, 1) == (int)gridDim.x - 1) {
      *a.integ.blocks_done = 0;
      const_cast<lb_ctrl*>(a.ctrl)->step = step + 1;
    }
  }
}

int lbk_decoder16(lb_engine* e, lb_gns* g) {
  lb_dec16_args a{};
  a.ctrl = e->ctrl;
  a.n_rows = e->BN;
  a.nlat = e->nlat;
  a.w0 = e->f16x2 ? g->dec_w0_h : g->dec_w0_f;
  a.b0 = g->dec.b0;
  a.w1 = e->f16x2 ? g->dec_w1_h : g->dec_w1_f;
  a.b1 = g->dec.b1;
  a.unscale = e->f16x2 ? g->dec_unscale : 1.f;
  a.acc_out = e->acc;
  a.out_dim = g->desc.out_dim;
  const int64_t tiles = std::max<int64_t>(1, (e->BN + 15) / 16);
  const int grid = (int)std::min<int64_t>(512, (tiles + GD_WAVES - 1) / GD_WAVES);
  if (e->integ_job.on && g->desc.out_dim == e->g.dim) {  // rollout step: the integrator rides along
    a.integ = e->integ_job;
    e->integ_done = true;
  }
  if (e->f16x2)
    hipLaunchKernelGGL((k_decoder16<true>), dim3(grid), dim3(GD_THREADS), 0, e->stream, a, e->g);
  else
    hipLaunchKernelGGL((k_decoder16<false>), dim3(grid), dim3(GD_THREADS), 0, e->stream, a, e->g);
  LB_HIP(hipGetLastError());
  return LB_OK;
}

// ---------------------------------------------------------------------------------- model build
// Blob order (models/gns.py GNS.flatten): [embed] then per MLP, in module-creation order,
// linear_0 .. linear_{n-1} as (w [in][out], b [out]) and, if present, LayerNorm scale, offset.
int lb_gns_create_generic(lb_engine* e, const lb_gns_desc* d, const float* w, int64_t n_floats, lb_gns** out) {
  const int D = LB_D, L = d->num_mp_steps, nl = d->blocks_per_step, dl = d->latent_size;
  if (nl < 1 || nl > 16) return lb_fail(LB_ERR_ARG, "num_mlp_layers %d out of range (1..16)", nl);
  const bool has_emb = d->num_particle_types > 1;
  const int emb = has_emb ? d->embedding_size : 0;
  const int nin = d->node_in + emb;
  if (nin > 128) return lb_fail(LB_ERR_UNSUPPORTED, "node input width %d > 128 not built", nin);
  const int kpad = (nin + 31) / 32 * 32;  // 32 .. 128

  std::vector<float> host;
  auto put = [&](const float* src, size_t n) -> size_t {
    size_t off = (host.size() + 63) & ~(size_t)63;
    host.resize(off + n, 0.f);
    if (src) memcpy(host.data() + off, src, n * sizeof(float));
    return off;
  };
  struct LinOff {
    std::vector<size_t> wh, wf;
    size_t b;
  };
  struct MlpOff {
    std::vector<LinOff> lin;
    size_t lns = 0, lno = 0;
    bool ln = false;
  };
  const float* p = w;
  const float* p_end = w + n_floats;
  bool short_blob = false;
  double w_rms_min = 1e30;  // see lb_gns_create: uniformly small weight matrices fall out of the f16x2 accuracy class
  auto note_rms = [&](const float* m, size_t n) {
    double s2 = 0;
    size_t nz = 0;
    for (size_t i = 0; i < n; ++i) {
      s2 += (double)m[i] * m[i];
      nz += m[i] != 0.f;
    }
    if (nz) w_rms_min = std::min(w_rms_min, std::sqrt(s2 / (double)nz));
  };
  // one Linear: in_blocks x blk_in input rows (each block padded to 128 k's), `outw` columns (padded to 128)
  auto read_linear = [&](int in_blocks, int blk_in, int outw) -> LinOff {
    LinOff o;
    const size_t need = (size_t)in_blocks * blk_in * outw + outw;
    if (p + need > p_end) {
      short_blob = true;
      o.b = 0;
      return o;
    }
    note_rms(p, (size_t)in_blocks * blk_in * outw);
    std::vector<float> tmp((size_t)128 * 128);
    for (int b = 0; b < in_blocks; ++b) {
      const float* src = p + (size_t)b * blk_in * outw;
      lb_pack_weight16h(src, blk_in, outw, 128, tmp.data(), 128);
      o.wh.push_back(put(tmp.data(), tmp.size()));
      lb_pack_weight16(src, blk_in, outw, 128, tmp.data());
      o.wf.push_back(put(tmp.data(), tmp.size()));
    }
    p += (size_t)in_blocks * blk_in * outw;
    std::vector<float> bias(128, 0.f);
    memcpy(bias.data(), p, sizeof(float) * outw);
    o.b = put(bias.data(), 128);
    p += outw;
    return o;
  };
  auto read_mlp = [&](int in_blocks, int blk_in, int outw, bool ln) -> MlpOff {
    MlpOff m;
    for (int li = 0; li < nl; ++li) {
      const bool first = li == 0, last = li == nl - 1;
      m.lin.push_back(read_linear(first ? in_blocks : 1, first ? blk_in : dl, last ? outw : dl));
      if (short_blob) return m;
    }
    m.ln = ln;
    if (ln) {
      if (p + 2 * (size_t)outw > p_end) {
        short_blob = true;
        return m;
      }
      std::vector<float> v(128, 0.f);
      memcpy(v.data(), p, sizeof(float) * outw);
      m.lns = put(v.data(), 128);
      p += outw;
      std::fill(v.begin(), v.end(), 0.f);
      memcpy(v.data(), p, sizeof(float) * outw);
      m.lno = put(v.data(), 128);
      p += outw;
    }
    return m;
  };
  size_t off_embed = 0;
  if (has_emb) {
    const size_t n = (size_t)d->num_particle_types * emb;
    if (p + n > p_end) return lb_fail(LB_ERR_ARG, "weight blob too short (embedding)");
    off_embed = put(p, n);
    p += n;
  }
  MlpOff o_en = read_mlp(1, nin, dl, true);
  MlpOff o_ee = read_mlp(1, d->edge_in, dl, true);
  std::vector<MlpOff> o_pe, o_pn;
  for (int k = 0; k < L && !short_blob; ++k) {
    o_pe.push_back(read_mlp(3, dl, dl, true));
    o_pn.push_back(read_mlp(2, dl, dl, true));
  }
  MlpOff o_dec;
  if (!short_blob) o_dec = read_mlp(1, dl, d->out_dim, false);
  if (short_blob || p != p_end)
    return lb_fail(LB_ERR_ARG, "weight blob has %lld floats, the model (num_mlp_layers %d, latent %d) needs %s",
                   (long long)n_floats, nl, dl, short_blob ? "more" : "fewer");

  lb_gns* g = new lb_gns();
  g->desc = *d;
  g->eng = e;
  g->tap = nullptr;
  g->generic = true;
  g->kq_node = kpad / 8;
  if (hipMalloc((void**)&g->blob, host.size() * sizeof(float)) != hipSuccess) {
    delete g;
    return lb_fail(LB_ERR_HIP, "hipMalloc(weights) failed");
  }
  if (hipMemcpy(g->blob, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
    lb_gns_destroy(g);
    return lb_fail(LB_ERR_HIP, "weight upload failed");
  }
  auto mk = [&](const MlpOff& m) {
    lb_gen_mlp r;
    for (const LinOff& lo : m.lin) {
      lb_gen_lin x;
      for (size_t o : lo.wh) x.wh.push_back(g->blob + o);
      for (size_t o : lo.wf) x.wf.push_back(g->blob + o);
      x.b = g->blob + lo.b;
      r.lin.push_back(x);
    }
    r.ln_s = m.ln ? g->blob + m.lns : nullptr;
    r.ln_o = m.ln ? g->blob + m.lno : nullptr;
    return r;
  };
  g->embed = has_emb ? g->blob + off_embed : nullptr;
  g->g_enc_node = mk(o_en);
  g->g_enc_edge = mk(o_ee);
  g->g_dec = mk(o_dec);
  for (int k = 0; k < L; ++k) {
    g->g_proc_edge.push_back(mk(o_pe[k]));
    g->g_proc_node.push_back(mk(o_pn[k]));
  }
  g->lnc[0] = 1.0f / (float)dl;  // LayerNorm width of this model (lb_gns_bind)
  g->lnc[1] = (float)(D - dl);
  if (w_rms_min < 0.0078125 && e->f16x2 && e->math_auto) {
    fprintf(stderr, "[lbhip] a weight matrix has rms %.3g < 2^-7: its fp16 hi/lo split would fall short of the 1e-5 class - "
                    "this engine uses exact-fp32 MFMA arithmetic\n", w_rms_min);
    e->f16x2 = 0;
  }
  const int64_t BN = e->BN;
  int rc = lb_ensure_node_scratch(e);
  if (!rc) rc = lb_gns_bind(e, g);
  for (int i = 0; i < 3 && !rc; ++i) rc = lb_alloc(&g->gen_hn[i], (size_t)BN * D);
  if (rc) {
    lb_gns_destroy(g);
    return rc;
  }
  *out = g;
  return LB_OK;
}

// -------------------------------------------------------------------------------------- forward
int lbk_gns_forward_generic(lb_engine* e, lb_gns* g) {
  hipStream_t s = e->stream;
  const int64_t BN = e->BN;
  const int D = LB_D, L = g->desc.num_mp_steps, nl = g->desc.blocks_per_step;
  int rc;
  // edge-sized hidden buffer, grown with the edge capacity (tile-blocked like the edge latents)
  if (nl > 1 && g->gen_he_cap < e->e_alloc) {
    LB_HIP(hipStreamSynchronize(s));
    if (g->gen_he) (void)hipFree(g->gen_he);
    g->gen_he = nullptr;
    g->gen_he_cap = 0;
    LB_TRY(lb_alloc(&g->gen_he, (size_t)(e->e_alloc + 32) * D));
    g->gen_he_cap = e->e_alloc;
  }
  const int64_t e_bound = e->e_alloc;
  auto W = [&](const lb_gen_lin& l, int blk) { return e->f16x2 ? l.wh[blk] : l.wf[blk]; };
  auto base_args = [&]() {
    lb_dense_args a{};
    a.ctrl = e->ctrl;
    a.x_stride = D;
    a.x_width = D;
    a.out_dim = g->desc.out_dim;
    return a;
  };
  // Linears 1..nl-1 of an MLP over `hid` (in place; hidden rows are stored before their ReLU); the
  // last one applies LayerNorm / residual / outputs as described by `fin`
  auto tail = [&](const lb_gen_mlp& m, float* hid, int blocked, int64_t n_rows, int64_t bound,
                  lb_dense_args fin) -> int {
    for (int li = 1; li < nl; ++li) {
      const bool last = li == nl - 1;
      lb_dense_args a = last ? fin : base_args();
      a.n_rows = n_rows;
      a.x = hid;
      a.x_blocked = blocked;
      a.w = W(m.lin[li], 0);
      a.bias = m.lin[li].b;
      if (!last) {
        a.out = hid;
        a.out_blocked = blocked;
      } else {
        a.ln_s = m.ln_s;
        a.ln_o = m.ln_o;
      }
      int r = lbk_dense16(e, a, bound, true);
      if (r) return r;
    }
    return LB_OK;
  };

  lb_tic(e, LB_T_NODEFEAT);
  rc = lbk_node_features(e, e->xnode, g->embed, g->desc.embedding_size, g->desc.num_particle_types, nullptr,
                         nullptr, nullptr, nullptr);
  lb_toc(e);
  if (rc) return rc;

  // ---- encoders (gns.py:65-84)
  lb_tic(e, LB_T_ENC_NODE);
  {
    lb_dense_args fin = base_args();
    fin.out = e->nlat;
    lb_dense_args a = nl == 1 ? fin : base_args();
    a.n_rows = BN;
    a.x = e->xnode;
    a.x_stride = e->g.kpad;
    a.x_width = e->g.kpad;
    a.w = W(g->g_enc_node.lin[0], 0);
    a.bias = g->g_enc_node.lin[0].b;
    if (nl == 1) {
      a.ln_s = g->g_enc_node.ln_s;
      a.ln_o = g->g_enc_node.ln_o;
    } else {
      a.out = g->gen_hn[1];
    }
    rc = lbk_dense16(e, a, BN, false);
    if (!rc) rc = tail(g->g_enc_node, g->gen_hn[1], 0, BN, BN, fin);
  }
  lb_toc(e);
  if (rc) return rc;
  lb_tic(e, LB_T_ENC_EDGE);
  {
    lb_dense_args fin = base_args();
    fin.out = e->elat;
    fin.out_blocked = 1;
    lb_dense_args a = nl == 1 ? fin : base_args();
    a.n_rows = -1;
    a.x = e->efeat;
    a.x_stride = 8;
    a.x_width = 8;
    a.w = W(g->g_enc_edge.lin[0], 0);
    a.bias = g->g_enc_edge.lin[0].b;
    if (nl == 1) {
      a.ln_s = g->g_enc_edge.ln_s;
      a.ln_o = g->g_enc_edge.ln_o;
    } else {
      a.out = g->gen_he;
      a.out_blocked = 1;
    }
    rc = lbk_dense16(e, a, e_bound, false);
    if (!rc) rc = tail(g->g_enc_edge, g->gen_he, 1, -1, e_bound, fin);
  }
  lb_toc(e);
  if (rc) return rc;
  if (g->tap) LB_HIP(hipMemcpyAsync(g->tap, e->nlat, sizeof(float) * BN * D, hipMemcpyDeviceToDevice, s));

  // ---- processor (gns.py:86-122)
  for (int k = 0; k < L; ++k) {
    const lb_gen_mlp& me = g->g_proc_edge[k];
    const lb_gen_mlp& mn = g->g_proc_node[k];
    lb_tic(e, LB_T_EDGE_MLP);
    {
      // sender / receiver blocks of the first Linear on the nodes (no bias): Ps -> hn[0], Pr -> hn[2]
      for (int blk = 0; blk < 2 && !rc; ++blk) {
        lb_dense_args a = base_args();
        a.n_rows = BN;
        a.x = e->nlat;
        a.w = W(me.lin[0], blk);
        a.out = g->gen_hn[blk == 0 ? 0 : 2];
        rc = lbk_dense16(e, a, BN, false);
      }
      lb_dense_args fin = base_args();
      fin.resid = e->elat;  // e' + e, gns.py:120-122
      fin.resid_blocked = 1;
      fin.out = e->elat;
      fin.out_blocked = 1;
      fin.out_pre = e->msg;  // the message that is aggregated is the MLP output itself
      lb_dense_args a = nl == 1 ? fin : base_args();
      a.n_rows = -1;
      a.x = e->elat;
      a.x_blocked = 1;
      a.w = W(me.lin[0], 2);
      a.bias = me.lin[0].b;
      a.add0 = g->gen_hn[0];
      a.idx0 = e->senders;
      a.add1 = g->gen_hn[2];
      a.idx1 = e->receivers;
      if (nl == 1) {
        a.ln_s = me.ln_s;
        a.ln_o = me.ln_o;
      } else {
        a.out = g->gen_he;
        a.out_blocked = 1;
      }
      if (!rc) rc = lbk_dense16(e, a, e_bound, false);
      if (!rc) rc = tail(me, g->gen_he, 1, -1, e_bound, fin);
    }
    lb_toc(e);
    if (rc) return rc;
    lb_tic(e, LB_T_AGGREGATE);
    rc = lbk_segment_sum(e, e->msg, e->agg, D);
    lb_toc(e);
    if (rc) return rc;
    lb_tic(e, LB_T_NODE_MLP);
    {
      lb_dense_args t0 = base_args();  // agg block of the first Linear -> hn[0]
      t0.n_rows = BN;
      t0.x = e->agg;
      t0.w = W(mn.lin[0], 1);
      t0.out = g->gen_hn[0];
      rc = lbk_dense16(e, t0, BN, false);
      lb_dense_args fin = base_args();
      fin.resid = e->nlat;  // n' + n
      fin.out = e->nlat;
      lb_dense_args a = nl == 1 ? fin : base_args();
      a.n_rows = BN;
      a.x = e->nlat;
      a.w = W(mn.lin[0], 0);
      a.bias = mn.lin[0].b;
      a.add0 = g->gen_hn[0];
      if (nl == 1) {
        a.ln_s = mn.ln_s;
        a.ln_o = mn.ln_o;
      } else {
        a.out = g->gen_hn[1];
      }
      if (!rc) rc = lbk_dense16(e, a, BN, false);
      if (!rc) rc = tail(mn, g->gen_hn[1], 0, BN, BN, fin);
    }
    lb_toc(e);
    if (rc) return rc;
    if (g->tap)
      LB_HIP(hipMemcpyAsync(g->tap + (size_t)(k + 1) * BN * D, e->nlat, sizeof(float) * BN * D,
                            hipMemcpyDeviceToDevice, s));
  }

  // ---- decoder (gns.py:125-133): no LayerNorm, out_dim columns
  lb_tic(e, LB_T_DECODER);
  {
    lb_dense_args fin = base_args();
    fin.acc_out = e->acc;
    lb_dense_args a = nl == 1 ? fin : base_args();
    a.n_rows = BN;
    a.x = e->nlat;
    a.w = W(g->g_dec.lin[0], 0);
    a.bias = g->g_dec.lin[0].b;
    if (nl > 1) a.out = g->gen_hn[1];
    rc = lbk_dense16(e, a, BN, false);
    if (!rc) rc = tail(g->g_dec, g->gen_hn[1], 0, BN, BN, fin);
  }
  lb_toc(e);
  return rc;
}
