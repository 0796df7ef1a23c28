// lb_lin32.h - the training step's tall-skinny products without the library: exact fp32 (k_lin32, k_lin32f) and, at the end of
// the file, the scaled f16x2 form the step uses by default (k_lin32h, k_pack_wh) (round 5; device code, included by
// lb_train.hip and by tools/lin_bench.hip):  Y[rows x NO] = X[rows x NR] * Wop[NR x NO]  with the elementwise neighbours of
// the product in the epilogue.  Wop = W (Y = X W: forward) or W^T (dX = dY W^T: backward); rows ~1e4 .. 1e5, NR, NO <= 256.
// Reference: the Linear layers of hk.nets.MLP inside GNS (models/gns.py:65-171) under value_and_grad (train/trainer.py:63-89).
//
// rocBLAS ran these at 60 - 94 TFLOP/s and every elementwise pass behind them (bias + ReLU, ReLU mask, the "+=" of a gradient
// with two producers) was a launch of its own over the same E x 128 array.
//
// Layout.  One wave per 16-row tile on v_mfma_f32_16x16x4_f32 (exact fp32 products, fp32 accumulate), 8 waves per workgroup,
// ONE workgroup per CU (2 waves per SIMD, 256 VGPRs each), the whole operand matrix in LDS in MFMA A-fragment order:
// f32x4 entry ((j * NOB + mb) * 64 + lane) holds, for i = 0 .. 3, Wop[16 j + 4 (lane >> 4) + i][16 mb + (lane & 15)] - one
// ds_read_b128 feeds the four k-steps (j, i) of output block mb.  The fragment order is produced ONCE per training step for all
// weight matrices by k_pack_w (both orientations; the weights only change in the optimiser): the workgroups then stage LDS with
// a flat, fully vectorised copy (a first version built the fragments inside k_lin32 from the row-major matrix with scalar
// loads: 512 workgroups x 32 dependent L2 round trips to the same 64 KiB - 10 us of a 40 us launch, 55 us on edge-sized calls
// where rocBLAS took 29).
// The B operand comes straight from global memory: lane (n, kq) loads X[row n][16 j + 4 kq .. + 3] (16 bytes) and uses element
// i in k-step (j, i) - the four k of an MFMA are then {16 j + 4 kq + i}, a permutation the A fragments follow.
// k_lin32f (the shapes that matter: NR a multiple of 128, NO = 128) keeps a RING of eight such loads in flight per lane -
// the fragment j + 8 of the sequence (tile, j), crossing into the wave's next tile - and fetches what the epilogue reads
// (ReLU mask, previous Y) at the top of the tile: 8 - 16 KiB in flight per wave, so the HBM side (64 - 96 MB per edge-sized
// call) overlaps the 256 MFMAs of a tile instead of being waited for one k-group at a time.
// D layout: lane (n, g) holds out[row n][16 mb + 4 g + jj]: one 16-byte store per block and lane.
// Epilogue (all optional): + bias[m]; ReLU; * (mask[row][m] > 0) (ReLU backward); += the previous Y (beta = 1).
// Every sum has a fixed order: bit-reproducible, and independent of which kernel (k_lin32 / k_lin32f) or grid ran it.
#pragma once
#include "lb_device.h"
#include "lb_f16x2.h"

struct lb_lin_args {
  const float* X;
  int ldx, NR;
  const float* Wp;      // the operand in fragment order (k_pack_w): NJ * NOB * 64 f32x4
  int NJ;               // ceil(NR / 16)
  float* Y;
  int ldy, NO;
  int64_t rows;
  const float* bias;    // [NO] or null
  int relu;
  const float* mask;    // [rows][ldm] or null: y *= (mask > 0)
  int ldm;
  int accum;            // y += Y
  // LayerNorm epilogue (k_lin32f<3>): Y receives z = X Wop WITHOUT the bias (what the backward reads), then
  // x = z + bias, y = scale * (x - mean) / sqrt(var + 1e-5) + offset over the first ln_d of the 128 columns, Yln = y (if not
  // null), Y2 = resid + y (if not null).  Row stride 128 for Yln / Y2 / resid.  (ln_d < 128: latents narrower than the
  // 128-wide rows - weights, biases and LayerNorm parameters are zero in the padded columns, so those columns of x are 0:
  // mean = sum / d, variance = (sum_128 (x - mean)^2 - (128 - d) mean^2) / d, padded outputs = scale 0 x .. + offset 0 = 0.)
  const float *ln_scale, *ln_offset, *resid;
  float *Yln, *Y2;
  int ln_d;
  // gather epilogue (k_lin32f<4>): y = relu(((X Wop + gat1[gidx1[row]]) + gat2[gidx2[row]]) + bias), rows of 128 - the edge
  // block's first Linear, [n_s | n_r | e] W0 + b0 with the two node-sized products gathered (k_edge_pre until round 5)
  const float *gat1, *gat2;
  const int32_t *gidx1, *gidx2;
  // k_lin32h: 1 / (power-of-two scale the operand matrix was packed with), written by k_pack_wh
  const float* wsc;
  // k_lin32h: if not null, receives the largest |X| of every 16-row tile (the kernel finds the rows' maxima anyway): the
  // weight-gradient kernel that contracts the same X as its dY operand scales by it instead of scanning X again
  float* tmax;
  // k_lin32h: floats between the 128-column chunks of an X row (0 = 128: one contiguous row).  The stacked product
  // dn += [dPs | dPr] [Ws | Wr]^T reads its two chunks from two arrays BN * 128 floats apart.
  int64_t xcs;
};

struct lb_pack_ent {    // one operand matrix of k_pack_w
  int64_t src, dst;     // float offsets into the weight blob / the packed blob
  int NR, NO, ldw, trans;  // trans = 0: Wop[k][m] = W[k * ldw + m]; 1: Wop[k][m] = W[m * ldw + k]
                           // (k_pack_wh only) 2: Wop[k][m] = W[((k >> 7) * NO + m) * ldw + (k & 127)] - the transposes of
                           // NR / 128 consecutive NO x 128 row blocks stacked along k
  int NJ, NOB;
};

// weight blob -> fragment order, zero padded; grid (blocks, entries)
__global__ void __launch_bounds__(256) k_pack_w(const float* __restrict__ w, float* __restrict__ wp,
                                                const lb_pack_ent* __restrict__ tab) {
  const lb_pack_ent e = tab[blockIdx.y];
  const int total = e.NJ * e.NOB * 256;
  const float* src = w + e.src;
  float* dst = wp + e.dst;
  for (int idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
    const int i = idx & 3, ln = (idx >> 2) & 63, q = idx >> 8;
    const int mb = q % e.NOB, j = q / e.NOB;
    const int k = 16 * j + 4 * (ln >> 4) + i, m = 16 * mb + (ln & 15);
    float v = 0.f;
    if (k < e.NR && m < e.NO) v = e.trans ? src[(int64_t)m * e.ldw + k] : src[(int64_t)k * e.ldw + m];
    dst[idx] = v;
  }
}

// flat copy of the packed operand into LDS (n4 f32x4), four loads in flight per thread
__device__ __forceinline__ void lb_lin_stage(f32x4* sW, const float* Wp, int n4, int tid) {
  const f32x4* src = reinterpret_cast<const f32x4*>(Wp);
  int idx = tid;
  for (; idx + 3 * 512 < n4; idx += 4 * 512) {
    const f32x4 v0 = src[idx], v1 = src[idx + 512], v2 = src[idx + 1024], v3 = src[idx + 1536];
    sW[idx] = v0;
    sW[idx + 512] = v1;
    sW[idx + 1024] = v2;
    sW[idx + 1536] = v3;
  }
  for (; idx < n4; idx += 512) sW[idx] = src[idx];
}

// ---- generic shapes (any NR, NO <= 16 NOB, unaligned rows): one k-group of X ahead
template <int NOB>
__global__ void __launch_bounds__(512) k_lin32(lb_lin_args a) {
  extern __shared__ f32x4 sWl[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = lane & 15, kq = lane >> 4;
  const int NJ = a.NJ;
  lb_lin_stage(sWl, a.Wp, NJ * NOB * 64, tid);
  __syncthreads();
  const int64_t ntiles = (a.rows + 15) >> 4;
  const bool x_vec = (a.ldx & 3) == 0 && (a.NR & 15) == 0;            // whole 16-byte loads stay inside the row
  const bool y_vec = (a.ldy & 3) == 0 && (a.NO & 3) == 0;
  for (int64_t t = (int64_t)wave * gridDim.x + blockIdx.x; t < ntiles; t += (int64_t)gridDim.x * 8) {
    const int64_t row = t * 16 + n, rowc = row < a.rows ? row : a.rows - 1;
    const float* xr = a.X + rowc * a.ldx;
    f32x4 acc[NOB];
#pragma unroll
    for (int mb = 0; mb < NOB; ++mb) acc[mb] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto load_x = [&](int j) -> f32x4 {
      const int c = 16 * j + 4 * kq;
      if (x_vec) return *reinterpret_cast<const f32x4*>(xr + c);
      f32x4 v;
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = c + i < a.NR ? xr[c + i] : 0.f;
      return v;
    };
    f32x4 xn = load_x(0);
    for (int j = 0; j < NJ; ++j) {
      const f32x4 x = xn;
      if (j + 1 < NJ) xn = load_x(j + 1);
      f32x4 wv[NOB];
#pragma unroll
      for (int mb = 0; mb < NOB; ++mb) wv[mb] = sWl[(j * NOB + mb) * 64 + lane];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int mb = 0; mb < NOB; ++mb) acc[mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[mb][i], x[i], acc[mb], 0, 0, 0);
    }
    if (row < a.rows) {
#pragma unroll
      for (int mb = 0; mb < NOB; ++mb) {
        const int m0 = 16 * mb + 4 * kq;
        if (m0 >= a.NO) continue;
        f32x4 y = acc[mb];
        if (a.bias) {
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) y[jj] += m0 + jj < a.NO ? a.bias[m0 + jj] : 0.f;
        }
        if (a.relu) {
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) y[jj] = fmaxf(y[jj], 0.f);
        }
        if (a.mask) {
          const float* mr = a.mask + row * a.ldm + m0;
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) y[jj] = (m0 + jj < a.NO && mr[jj] > 0.f) ? y[jj] : 0.f;
        }
        float* yr = a.Y + row * a.ldy + m0;
        if (y_vec) {
          f32x4* y4 = reinterpret_cast<f32x4*>(yr);
          *y4 = a.accum ? *y4 + y : y;
        } else {
#pragma unroll
          for (int jj = 0; jj < 4; ++jj)
            if (m0 + jj < a.NO) yr[jj] = a.accum ? yr[jj] + y[jj] : y[jj];
        }
      }
    }
  }
}

// ---- the shapes that matter: NR % 128 == 0, NO == 128, ldx % 4 == 0, ldy % 4 == 0.
// EPI: 0 bias / ReLU, 1 ReLU mask, 2 accumulate, 3 LayerNorm (+ residual), 4 two gathered rows + bias + ReLU.
// Software pipeline (the compiler neither double-buffers the LDS reads nor keeps the global loads where they are written: it
// sinks all eight of a chunk behind the chunk's last MFMA, which exposes the latency of the first one): per k-group the LDS
// reads of the NEXT group are issued first, then the 32 MFMAs of this group, then the ring slot is refilled; sched_barriers pin
// the three parts.  Bias / LayerNorm scale / offset sit in LDS behind the operand (a global load in the epilogue would wait,
// through the in-order vmcnt, for the stores in front of it).
template <int EPI>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) k_lin32f(lb_lin_args a) {
  constexpr int NOB = 8;
  extern __shared__ f32x4 sWl[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = lane & 15, kq = lane >> 4;
  const int NJ = a.NJ, nch = NJ >> 3;
  const int64_t ntiles = (a.rows + 15) >> 4, tstep = (int64_t)gridDim.x * 8;
  int64_t t = (int64_t)wave * gridDim.x + blockIdx.x;
  auto row_ptr = [&](int64_t tt) -> const float* {
    int64_t r = tt * 16 + n;
    r = r < a.rows ? r : a.rows - 1;
    return a.X + r * a.ldx + 4 * kq;
  };
  // the first eight fragments are requested before the operand is staged
  const float* xr = row_ptr(t);
  f32x4 ring[8];
#pragma unroll
  for (int jj = 0; jj < 8; ++jj) ring[jj] = *reinterpret_cast<const f32x4*>(xr + 16 * jj);
  f32x4* sV = sWl + NJ * NOB * 64;  // [32] bias, [32] scale, [32] offset
  if (tid < 96) {
    const float* src = tid < 32 ? a.bias : (tid < 64 ? a.ln_scale : a.ln_offset);
    sV[tid] = (src && (EPI == 3 || tid < 32)) ? reinterpret_cast<const f32x4*>(src)[tid & 31] : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  lb_lin_stage(sWl, a.Wp, NJ * NOB * 64, tid);
  __syncthreads();
  const bool has_bias = a.bias != nullptr;
  int gi1 = 0, gi2 = 0;  // EPI 4: the gather indices of the wave's current tile (fetched one tile ahead)
  if (EPI == 4) {
    int64_t r = t * 16 + n;
    r = r < a.rows ? r : a.rows - 1;
    gi1 = a.gidx1[r];
    gi2 = a.gidx2[r];
  }
  const f32x4* sw0 = sWl + lane;
  const f32x4* sv = sV + kq;  // + 4 mb: columns 16 mb + 4 kq ..
  f32x4 wv[2][NOB];
#pragma unroll
  for (int mb = 0; mb < NOB; ++mb) wv[0][mb] = sw0[mb * 64];
  for (; t < ntiles; t += tstep) {
    const int64_t row = t * 16 + n;
    const bool live = row < a.rows;
    const int64_t rowc = live ? row : a.rows - 1;
    const float* xnext = row_ptr(t + tstep);
    f32x4 ep[NOB], ep2[EPI == 4 ? NOB : 1];
    if (EPI == 1) {
      const float* mr = a.mask + rowc * a.ldm + 4 * kq;
#pragma unroll
      for (int mb = 0; mb < NOB; ++mb) ep[mb] = *reinterpret_cast<const f32x4*>(mr + 16 * mb);
    } else if (EPI == 2) {
      const float* yo = a.Y + rowc * a.ldy + 4 * kq;
#pragma unroll
      for (int mb = 0; mb < NOB; ++mb) ep[mb] = *reinterpret_cast<const f32x4*>(yo + 16 * mb);
    } else if (EPI == 4) {
      const float *p1 = a.gat1 + (int64_t)gi1 * 128 + 4 * kq, *p2 = a.gat2 + (int64_t)gi2 * 128 + 4 * kq;
#pragma unroll
      for (int mb = 0; mb < NOB; ++mb) {
        ep[mb] = *reinterpret_cast<const f32x4*>(p1 + 16 * mb);
        ep2[mb] = *reinterpret_cast<const f32x4*>(p2 + 16 * mb);
      }
      int64_t rn = (t + tstep) * 16 + n;
      rn = rn < a.rows ? rn : a.rows - 1;
      gi1 = a.gidx1[rn];
      gi2 = a.gidx2[rn];
    } else if (EPI == 3) {
      if (a.resid) {
        const float* rr = a.resid + rowc * 128 + 4 * kq;
#pragma unroll
        for (int mb = 0; mb < NOB; ++mb) ep[mb] = *reinterpret_cast<const f32x4*>(rr + 16 * mb);
      }
    }
    f32x4 acc[NOB];
#pragma unroll
    for (int mb = 0; mb < NOB; ++mb) acc[mb] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int jc = 0; jc < nch; ++jc) {
      const bool more = jc + 1 < nch;
      const float* nx = more ? xr + 128 * (jc + 1) : xnext;
      const f32x4* sw = sw0 + (jc * 8 * NOB) * 64;
      const f32x4* swn = more ? sw + 8 * NOB * 64 : sw0;  // the k-group behind this chunk's last: next chunk / next tile
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) {
        const f32x4* sn = jj < 7 ? sw + (jj + 1) * NOB * 64 : swn;
#pragma unroll
        for (int mb = 0; mb < NOB; ++mb) wv[(jj + 1) & 1][mb] = sn[mb * 64];
        __builtin_amdgcn_sched_barrier(0);
        const f32x4 x = ring[jj];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int mb = 0; mb < NOB; ++mb)
            acc[mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[jj & 1][mb][i], x[i], acc[mb], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        ring[jj] = *reinterpret_cast<const f32x4*>(nx + 16 * jj);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (live) {
      float* yr = a.Y + row * a.ldy + 4 * kq;
      if (EPI == 3) {
        // z (without the bias) for the backward; LayerNorm of z + bias over the row: the lanes (n, 0 .. 3) hold it
        float s = 0.f;
#pragma unroll
        for (int mb = 0; mb < NOB; ++mb) {
          *reinterpret_cast<f32x4*>(yr + 16 * mb) = acc[mb];
          acc[mb] = acc[mb] + sv[4 * mb];
          s += (acc[mb][0] + acc[mb][1]) + (acc[mb][2] + acc[mb][3]);
        }
        s += __shfl_xor(s, 16);
        s += __shfl_xor(s, 32);
        const float inv_d = 1.f / (float)a.ln_d;
        const float mean = s * inv_d;
        float q = 0.f;
#pragma unroll
        for (int mb = 0; mb < NOB; ++mb) {
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) acc[mb][jj] -= mean;
          q += (acc[mb][0] * acc[mb][0] + acc[mb][1] * acc[mb][1]) + (acc[mb][2] * acc[mb][2] + acc[mb][3] * acc[mb][3]);
        }
        q += __shfl_xor(q, 16);
        q += __shfl_xor(q, 32);
        q -= (float)(128 - a.ln_d) * mean * mean;  // the padded columns hold 0 - mean
        const float rs = 1.0f / sqrtf(q * inv_d + 1e-5f);
        float* y1 = a.Yln ? a.Yln + row * 128 + 4 * kq : nullptr;
        float* y2 = a.Y2 ? a.Y2 + row * 128 + 4 * kq : nullptr;
#pragma unroll
        for (int mb = 0; mb < NOB; ++mb) {
          const f32x4 sc = sv[32 + 4 * mb], of = sv[64 + 4 * mb];
          f32x4 y;
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) y[jj] = sc[jj] * (acc[mb][jj] * rs) + of[jj];
          if (y1) *reinterpret_cast<f32x4*>(y1 + 16 * mb) = y;
          if (y2) *reinterpret_cast<f32x4*>(y2 + 16 * mb) = a.resid ? y + ep[mb] : y;
        }
      } else {
#pragma unroll
        for (int mb = 0; mb < NOB; ++mb) {
          f32x4 y = acc[mb];
          if (has_bias) y = y + sv[4 * mb];
          if (a.relu) {
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) y[jj] = fmaxf(y[jj], 0.f);
          }
          if (EPI == 1) {
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) y[jj] = ep[mb][jj] > 0.f ? y[jj] : 0.f;
          } else if (EPI == 2) {
            y = ep[mb] + y;
          }
          if (EPI == 4) {
            y = ((acc[mb] + ep[mb]) + ep2[mb]) + sv[4 * mb];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) y[jj] = fmaxf(y[jj], 0.f);
          }
          *reinterpret_cast<f32x4*>(yr + 16 * mb) = y;
        }
      }
    }
    xr = xnext;
  }
}


// ---------------------------------------------------------------------------------------------------------------------
// k_lin32h: the same product in f16x2 arithmetic - every fp32 operand is carried as an fp16 hi + lo pair on
// v_mfma_f32_16x16x32_f16 (hi hi + hi lo + lo hi, fp32 accumulate: 96 MFMAs of 16 cycles per 16 x 128 x 128 tile instead of 256
// of 32), which is what the inference kernels do (lb_edge16v.hip).  Inference protects the fp16 range with a guard and an fp32
// redo; a training step cannot redo half a backward pass, and its operands (gradients of 1e-6, post-ReLU rows of any size) are
// not O(1) - so the range is MADE safe instead: every 128-wide chunk of every row is multiplied by the power of two that puts
// its largest entry into [1, 2) before it is split (exact), the chunk's product is accumulated on its own and joins the row's
// fp32 total times the inverse power (exact); the operand matrix is packed times the power of two that puts ITS largest entry
// into [1, 2) (k_pack_wh).  The error of a product term is then <= 2^-22 of (row-chunk maximum x matrix maximum), with no
// dependence on the magnitudes themselves.  Layout, ring, epilogues: k_lin32f's.  LDS entry ((p * 8 + mb) * 2 + part) * 64 + lane
// holds the eight halves W[32 p + {4 g .. 4 g + 3, 16 + 4 g .. 16 + 4 g + 3}][16 mb + (lane & 15)], g = lane >> 4, part 0 = hi,
// 1 = lo (the k order inside a 32-block is free as long as both operands use it: this one makes the B operand of k-step p the
// ring fragments 2 p and 2 p + 1 as they are).
struct lb_pack_ent_h {  // one operand matrix of k_pack_wh
  int64_t src, dst;     // float offsets into the weight blob / the packed blob (NP * NOB * 512 floats)
  int NR, NO, ldw, trans;
  int NP, NOB;
  int64_t sc;           // float offset of this matrix' inverse scale in the scale array
};
__global__ void __launch_bounds__(256) k_pack_wh(const float* __restrict__ w, float* __restrict__ wp, float* __restrict__ wsc,
                                                 const lb_pack_ent_h* __restrict__ tab) {
  __shared__ float s_m[256];
  const lb_pack_ent_h e = tab[blockIdx.y];
  const float* src = w + e.src;
  // the matrix' largest magnitude (every block of the entry computes it for itself: <= 49 k elements)
  float m = 0.f;
  {   // in memory order (the maximum does not care which way the matrix is read): rows of ldw floats, `cols` used
    const int rows_m = e.trans == 2 ? (e.NR >> 7) * e.NO : (e.trans ? e.NO : e.NR);
    const int cols = e.trans == 2 ? 128 : (e.trans ? e.NR : e.NO);
    const int tpr = cols >= 256 ? 256 : (cols >= 128 ? 128 : (cols >= 64 ? 64 : 32)), rpi = 256 / tpr;
    const int c0 = threadIdx.x % tpr, r0 = threadIdx.x / tpr;
    for (int c = c0; c < cols; c += tpr) {
      int r = r0;
      for (; r + 7 * rpi < rows_m; r += 8 * rpi) {   // eight loads in flight
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = src[(int64_t)(r + u * rpi) * e.ldw + c];
#pragma unroll
        for (int u = 0; u < 8; ++u) m = fmaxf(m, fabsf(v[u]));
      }
      for (; r < rows_m; r += rpi) m = fmaxf(m, fabsf(src[(int64_t)r * e.ldw + c]));
    }
  }
  s_m[threadIdx.x] = m;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) s_m[threadIdx.x] = fmaxf(s_m[threadIdx.x], s_m[threadIdx.x + o]);
    __syncthreads();
  }
  m = s_m[0];
  unsigned ex = (__float_as_uint(m) >> 23) & 0xffu;
  ex = m == 0.f ? 127u : (ex < 1u ? 1u : (ex > 253u ? 253u : ex));
  const float s = __uint_as_float((254u - ex) << 23);
  if (blockIdx.x == 0 && threadIdx.x == 0) wsc[e.sc] = __uint_as_float(ex << 23);
  _Float16* dst = reinterpret_cast<_Float16*>(wp + e.dst);
  const int total = e.NP * e.NOB * 512;   // (p, mb, lane, i)
  for (int idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
    const int i = idx & 7, ln = (idx >> 3) & 63, q = idx >> 9;
    const int mb = q % e.NOB, p = q / e.NOB, g = ln >> 4;
    const int k = 32 * p + (i < 4 ? 4 * g + i : 16 + 4 * g + (i - 4)), c = 16 * mb + (ln & 15);
    float v = 0.f;
    if (k < e.NR && c < e.NO)
      v = (e.trans == 2 ? src[((int64_t)(k >> 7) * e.NO + c) * e.ldw + (k & 127)]
                        : (e.trans ? src[(int64_t)c * e.ldw + k] : src[(int64_t)k * e.ldw + c])) * s;
    const _Float16 hi = (_Float16)v;
    const _Float16 lo = (_Float16)(v - (float)hi);
    const size_t base = ((size_t)(p * e.NOB + mb) * 2) * 64;
    dst[(base + ln) * 8 + i] = hi;
    dst[(base + 64 + ln) * 8 + i] = lo;
  }
}

// Paired launches (round 6): gridDim.y = 2 runs two products of the same shape and epilogue class in ONE launch, job b on
// blockIdx.y = 1 (the sender / receiver projections of the edge block, the two halves of d [n | agg]): node-sized products
// are latency chains of ~6 us, and two of them side by side cost one.  EPI 2 honours the job's own `accum` flag, so an
// accumulating product pairs with a storing one.
template <int EPI>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) k_lin32h(lb_lin_args a_job, lb_lin_args b_job) {
  const lb_lin_args& a = blockIdx.y ? b_job : a_job;
  constexpr int NOB = 8;
  const int64_t xcs = a.xcs ? a.xcs : 128;
  extern __shared__ f32x4 sWl[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = lane & 15, kq = lane >> 4;
  const int NJ = a.NJ, nch = NJ >> 3;
  const int64_t ntiles = (a.rows + 15) >> 4, tstep = (int64_t)gridDim.x * 8;
  int64_t t = (int64_t)wave * gridDim.x + blockIdx.x;
  auto row_ptr = [&](int64_t tt) -> const float* {
    int64_t r = tt * 16 + n;
    r = r < a.rows ? r : a.rows - 1;
    return a.X + r * a.ldx + 4 * kq;
  };
  const float* xr = row_ptr(t);
  f32x4 ring[8];
#pragma unroll
  for (int jj = 0; jj < 8; ++jj) ring[jj] = *reinterpret_cast<const f32x4*>(xr + 16 * jj);
  f32x4* sV = sWl + NJ * NOB * 64;  // [32] bias, [32] scale, [32] offset
  if (tid < 96) {
    const float* src = tid < 32 ? a.bias : (tid < 64 ? a.ln_scale : a.ln_offset);
    sV[tid] = (src && (EPI == 3 || tid < 32)) ? reinterpret_cast<const f32x4*>(src)[tid & 31] : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  lb_lin_stage(sWl, a.Wp, NJ * NOB * 64, tid);
  const float winv = *a.wsc;
  __syncthreads();
  const bool has_bias = a.bias != nullptr;
  int gi1 = 0, gi2 = 0;
  if (EPI == 4) {
    int64_t r = t * 16 + n;
    r = r < a.rows ? r : a.rows - 1;
    gi1 = a.gidx1[r];
    gi2 = a.gidx2[r];
  }
  const h8* sw0 = reinterpret_cast<const h8*>(sWl) + lane;
  const f32x4* sv = sV + kq;
  for (; t < ntiles; t += tstep) {
    const int64_t row = t * 16 + n;
    const bool live = row < a.rows;
    const int64_t rowc = live ? row : a.rows - 1;
    const float* xnext = row_ptr(t + tstep);
    f32x4 ep[NOB], ep2[EPI == 4 ? NOB : 1];
    if (EPI == 1) {
      const float* mr = a.mask + rowc * a.ldm + 4 * kq;
#pragma unroll
      for (int mb = 0; mb < NOB; ++mb) ep[mb] = *reinterpret_cast<const f32x4*>(mr + 16 * mb);
    } else if (EPI == 2) {
      const float* yo = a.Y + rowc * a.ldy + 4 * kq;
#pragma unroll
      for (int mb = 0; mb < NOB; ++mb) ep[mb] = *reinterpret_cast<const f32x4*>(yo + 16 * mb);
    } else if (EPI == 4) {
      const float *p1 = a.gat1 + (int64_t)gi1 * 128 + 4 * kq, *p2 = a.gat2 + (int64_t)gi2 * 128 + 4 * kq;
#pragma unroll
      for (int mb = 0; mb < NOB; ++mb) {
        ep[mb] = *reinterpret_cast<const f32x4*>(p1 + 16 * mb);
        ep2[mb] = *reinterpret_cast<const f32x4*>(p2 + 16 * mb);
      }
      int64_t rn = (t + tstep) * 16 + n;
      rn = rn < a.rows ? rn : a.rows - 1;
      gi1 = a.gidx1[rn];
      gi2 = a.gidx2[rn];
    } else if (EPI == 3) {
      if (a.resid) {
        const float* rr = a.resid + rowc * 128 + 4 * kq;
#pragma unroll
        for (int mb = 0; mb < NOB; ++mb) ep[mb] = *reinterpret_cast<const f32x4*>(rr + 16 * mb);
      }
    }
    f32x4 acc[NOB];   // the row's total over its chunks, in true units
#pragma unroll
    for (int mb = 0; mb < NOB; ++mb) acc[mb] = f32x4{0.f, 0.f, 0.f, 0.f};
    float mrow = 0.f;
    for (int jc = 0; jc < nch; ++jc) {
      // the chunk's largest magnitude over the row (this lane's 32 values, then the four lanes of the row)
      float m = 0.f;
#pragma unroll
      for (int jj = 0; jj < 8; ++jj)
#pragma unroll
        for (int i = 0; i < 4; ++i) m = fmaxf(m, fabsf(ring[jj][i]));
      m = fmaxf(m, __shfl_xor(m, 16));
      m = fmaxf(m, __shfl_xor(m, 32));
      mrow = fmaxf(mrow, m);
      unsigned ex = (__float_as_uint(m) >> 23) & 0xffu;
      ex = m == 0.f ? 127u : (ex < 1u ? 1u : (ex > 253u ? 253u : ex));
      const float s = __uint_as_float((254u - ex) << 23), inv = __uint_as_float(ex << 23) * winv;
      h8 hi[4], lo[4];
#pragma unroll
      for (int p = 0; p < 4; ++p) lb_split8v(ring[2 * p] * s, ring[2 * p + 1] * s, hi[p], lo[p]);
      // the ring is free: the next chunk (or the next tile's first) is requested before the first MFMA of this one
      const float* nx = jc + 1 < nch ? xr + xcs * (jc + 1) : xnext;
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) ring[jj] = *reinterpret_cast<const f32x4*>(nx + 16 * jj);
      f32x4 part[NOB];
#pragma unroll
      for (int mb = 0; mb < NOB; ++mb) part[mb] = f32x4{0.f, 0.f, 0.f, 0.f};
      const h8* sw = sw0 + (size_t)(jc * 4) * NOB * 2 * 64;
      // groups of (k-step p, GM column blocks): the 2 GM 16-byte LDS reads of group g + 1 are issued before the 3 GM MFMAs of
      // group g (left to itself the compiler reads a fragment pair right before its three MFMAs and waits for it).  GM = 4
      // where the registers allow it (two waves per SIMD: 256), 2 for the gather epilogue, which holds two more rows.
      constexpr int GM = EPI == 4 ? 2 : 4, NG = 4 * NOB / GM;
      h8 wq[2][2 * GM];
#pragma unroll
      for (int u = 0; u < 2 * GM; ++u) wq[0][u] = sw[u * 64];
#pragma unroll
      for (int gq = 0; gq < NG; ++gq) {
        const int p = gq / (NOB / GM), m0 = (gq % (NOB / GM)) * GM;
        if (gq + 1 < NG) {
#pragma unroll
          for (int u = 0; u < 2 * GM; ++u) wq[(gq + 1) & 1][u] = sw[(((gq + 1) * GM) * 2 + u) * 64];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < GM; ++q) {
          const h8 wh = wq[gq & 1][2 * q], wl = wq[gq & 1][2 * q + 1];
          part[m0 + q] = MFMA16H(wh, hi[p], part[m0 + q]);
          part[m0 + q] = MFMA16H(wh, lo[p], part[m0 + q]);
          part[m0 + q] = MFMA16H(wl, hi[p], part[m0 + q]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int mb = 0; mb < NOB; ++mb)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) acc[mb][jj] += part[mb][jj] * inv;
    }
    if (a.tmax) {   // the tile's largest |X| (rows past the end repeat the last row)
#pragma unroll
      for (int off = 1; off < 16; off <<= 1) mrow = fmaxf(mrow, __shfl_xor(mrow, off));
      if (lane == 0) a.tmax[t] = mrow;
    }
    if (live) {
      float* yr = a.Y + row * a.ldy + 4 * kq;
      if (EPI == 3) {
        float s = 0.f;
#pragma unroll
        for (int mb = 0; mb < NOB; ++mb) {
          *reinterpret_cast<f32x4*>(yr + 16 * mb) = acc[mb];
          acc[mb] = acc[mb] + sv[4 * mb];
          s += (acc[mb][0] + acc[mb][1]) + (acc[mb][2] + acc[mb][3]);
        }
        s += __shfl_xor(s, 16);
        s += __shfl_xor(s, 32);
        const float inv_d = 1.f / (float)a.ln_d;
        const float mean = s * inv_d;
        float q = 0.f;
#pragma unroll
        for (int mb = 0; mb < NOB; ++mb) {
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) acc[mb][jj] -= mean;
          q += (acc[mb][0] * acc[mb][0] + acc[mb][1] * acc[mb][1]) + (acc[mb][2] * acc[mb][2] + acc[mb][3] * acc[mb][3]);
        }
        q += __shfl_xor(q, 16);
        q += __shfl_xor(q, 32);
        q -= (float)(128 - a.ln_d) * mean * mean;
        const float rs = 1.0f / sqrtf(q * inv_d + 1e-5f);
        float* y1 = a.Yln ? a.Yln + row * 128 + 4 * kq : nullptr;
        float* y2 = a.Y2 ? a.Y2 + row * 128 + 4 * kq : nullptr;
#pragma unroll
        for (int mb = 0; mb < NOB; ++mb) {
          const f32x4 sc = sv[32 + 4 * mb], of = sv[64 + 4 * mb];
          f32x4 y;
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) y[jj] = sc[jj] * (acc[mb][jj] * rs) + of[jj];
          if (y1) *reinterpret_cast<f32x4*>(y1 + 16 * mb) = y;
          if (y2) *reinterpret_cast<f32x4*>(y2 + 16 * mb) = a.resid ? y + ep[mb] : y;
        }
      } else {
#pragma unroll
        for (int mb = 0; mb < NOB; ++mb) {
          f32x4 y = acc[mb];
          if (has_bias) y = y + sv[4 * mb];
          if (a.relu) {
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) y[jj] = fmaxf(y[jj], 0.f);
          }
          if (EPI == 1) {
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) y[jj] = ep[mb][jj] > 0.f ? y[jj] : 0.f;
          } else if (EPI == 2) {
            if (a.accum) y = ep[mb] + y;
          }
          if (EPI == 4) {
            y = ((acc[mb] + ep[mb]) + ep2[mb]) + sv[4 * mb];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) y[jj] = fmaxf(y[jj], 0.f);
          }
          *reinterpret_cast<f32x4*>(yr + 16 * mb) = y;
        }
      }
    }
    xr = xnext;
  }
}
