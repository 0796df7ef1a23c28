// lb_train.hip - the GNS training step on the device: forward with saved activations, MSE loss, hand-written
// backward, fused AdamW.
//
// Reference: lagrangebench/train/trainer.py:35-60 (_mse: weighted squared error of the predicted normalised
// accelerations, summed over dim, masked to the non-kinematic particles, divided by their number) and :63-89
// (_update: vmap(value_and_grad) over the batch, gradients SUMMED over the batch, loss averaged, optax.adamw);
// the network is GNS.__call__ of models/gns.py:65-171 (haiku MLP / LayerNorm / Embed, jraph GraphNetwork).
//
// Design.  Training is the row next to the hot path, not the hot path.  Round 5: no library GEMM is left.  The two tall-skinny
// products of every Linear - Y = X W and dX = dY W^T, ~1e5 rows against a <= 256 x 128 matrix - run on k_lin32f (lb_lin32.h:
// one wave per 16-row tile, v_mfma_f32_16x16x4_f32, the operand matrix in LDS in fragment order, packed once per step by
// k_pack_w) with their elementwise neighbours in the epilogue (bias + ReLU forward, ReLU mask backward, "+=" for gradients
// with two producers); rounds 3 - 4 sent them to rocBLAS sgemm (dlopen-ed) and paid a launch per elementwise pass.  The THIRD
// product, dW += X^T dY, is a reduction over ~1e5 rows into a tiny result, which the library ran at 16 TFLOP/s: hand-written
// since round 4 (k_dw_part: fp32 MFMA, split over the rows, bias column sums folded in, partials combined in a fixed order -
// 95 TFLOP/s on the 384 x 128 case; k_dw_narrow for the decoder's 128 x dim matrix).
// Round 5, second half: by default all three products run in f16x2 arithmetic - fp16 hi / lo pairs on v_mfma_f32_16x16x32_f16,
// the operands put into fp16's range by exact power-of-two scales per row chunk / matrix / row block instead of a guard
// (k_lin32h in lb_lin32.h, k_dw_part_h below; LB_TRAIN_MATH=f32 keeps the exact kernels): 6.05 -> 4.5 ms per TGV3D-8k step.
// The edge block never forms [n_s | n_r | e]: its first Linear is split by rows of W0 into two node-sized products and one
// edge-sized one (the gather epilogue of k_lin32f / k_edge_dP).  Hand-written HIP for everything that is not a GEMM: [n | agg], bias + ReLU,
// LayerNorm forward and backward (the backward keeps the running sums of d scale / d offset in registers, no scratch copy),
// ReLU masking, jraph.segment_sum and its transpose on the receiver-sorted CSR, the sender-side transpose of the gather
// through a sender-sorted permutation (stable radix sort: ascending edge order per sender, no atomics), the embedding-table
// gradient, the masked MSE and its gradient, AdamW over the flat parameter blob.  Every floating-point sum has a fixed
// order: two runs of a step produce the same bits.  A batch of B trajectories is one disjoint graph: the per-trajectory
// losses of _update (gradients summed, loss averaged) are obtained in one pass with a per-node weight 1 / n_nonkinematic(b).
// The graph (receiver-sorted edge list), node / edge features and normalisation are the engine's: the caller runs
// case.preprocess (noise, neighbor list, features, targets) first, exactly as for inference.
// Weights live in fp32 in the layout of GNS.flatten (= lb_gns_create's blob): [embed] then per MLP w0 (in x 128),
// b0, w1 (128 x out), b1 [, LayerNorm scale, offset]; gradients and both AdamW moments use the same layout.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

#include "lb_device.h"
#include "lb_lin32.h"

#include <hipcub/hipcub.hpp>  // header-only device radix sort (the sender-sorted edge permutation of the gather's transpose)

#define TD 128  // latent width of the training path (GNS-*-128)

struct lb_train_mlp {
  int64_t w0, b0, w1, b1, lns, lno;  // float offsets into the blob: first Linear, LAST Linear, LayerNorm
  std::vector<int64_t> wm, bm;       // the num_mlp_layers - 2 Linears between them (128 x 128 each; round 6)
  int in, out;
  bool ln;
};

// one gradient reduction of the step (k_part_reduce)
struct lb_red_ent {
  int64_t part;     // float offset of the slot in the partial buffer
  int64_t stride;   // floats between the partials of consecutive g
  int64_t dst0, dst1;  // float offsets into the gradient blob (dst1 unused when n1 == 0)
  int G, n0, n1, off1;
  int blk0, pad;    // first block of this reduction in the flat grid (RED_OUT = 256 outputs per block)
};

// LB_TRAIN_MATH=f32: the exact-fp32 product kernels (k_lin32f); default: f16x2 (k_lin32h)
#define LB_DW_CALLS 512   // k_dw_part_h launches per step the X range bookkeeping covers (GNS-10: 62, SEGNN-10-64: ~90)
static bool lb_train_f16x2_default() {
  const char* m = getenv("LB_TRAIN_MATH");
  return !(m && (m[0] == 'f' && m[1] == '3'));
}

struct lb_sgt;  // SEGNN-specific state of a training handle (lb_train_segnn.h)

struct lb_gns_train {
  lb_sgt* sg = nullptr;   // non-null: this handle trains a SEGNN (created by lb_segnn_train_create)
  lb_gns_desc desc;
  lb_engine* eng;
  int64_t n_floats = 0;   // floats of the DEVICE blobs (latent padded to 128)
  int64_t n_compact = 0;  // floats of the caller's blob (GNS.flatten with the model's latent size)
  int lat = TD;           // the model's latent size (<= 128)
  std::vector<int64_t> cmap;  // caller index -> device index (empty: identity, latent 128)
  float *w = nullptr, *g = nullptr, *m = nullptr, *v = nullptr;  // weights, gradients, AdamW moments
  int64_t off_embed = 0;
  lb_train_mlp enc_node, enc_edge, dec;
  std::vector<lb_train_mlp> pe, pn;
  int nin = 0, kpad = 0;
  int64_t step = 0;
  // activations (sized for cap_n nodes / cap_e edges)
  int64_t cap_n = 0, cap_e = 0;
  float *xnode = nullptr, *a_en = nullptr, *z_en = nullptr, *a_ee = nullptr, *z_ee = nullptr;
  std::vector<float*> nlat, elat;            // [L+1] node / edge latents entering layer k
  std::vector<float*> ae, ze, xn, an, zn;  // per layer: post-ReLU hidden, pre-LayerNorm output; node block: concat input
  float *a_d = nullptr, *pred = nullptr;
  float *dn = nullptr, *de = nullptr, *dy = nullptr, *dz = nullptr, *da = nullptr, *dx = nullptr, *dagg = nullptr;
  float* agg = nullptr;
  // partial sums of every gradient reduction of a step (k_dw_part / k_dw_narrow / k_ln_bwd2 / k_colsum_part): one slot per
  // producer, summed in one k_part_reduce launch at the end of the backward pass
  float* dwpart = nullptr;
  int64_t red_cap = 0, red_off = 0;
  std::vector<lb_red_ent> red_tab;
  lb_red_ent *red_host = nullptr, *red_dev = nullptr;  // pinned staging copy / device table
  int red_blocks = 0;
  float* proj = nullptr;     // edge block: [n W_s ; n W_r] (2 x cap_n x 128), reused for their gradients
  float* node_w = nullptr;   // per node loss weight (0 for kinematic particles)
  double* loss_dev = nullptr;
  double* loss_part = nullptr;  // one partial per wave of k_mse_grad
  int32_t* cnt_dev = nullptr;  // non-kinematic particles per trajectory
  // sender-sorted view of the edge list (round 4): the transpose of the [n_s | n_r | e] gather sums, per node, the
  // gradient rows of the edges it SENDS - in ascending edge order, no atomics, bit-reproducible
  int32_t *snd_key = nullptr, *snd_perm = nullptr, *iota = nullptr, *snd_key_in = nullptr, *snd_ptr = nullptr;
  void* sort_tmp = nullptr;
  size_t sort_tmp_bytes = 0;
  int64_t sort_cap = 0;
  // the operand matrices of k_lin32 in MFMA fragment order (lb_lin32.h): registered at their first use, re-packed from the
  // weight blob at the top of every lb_gns_train_loss_grad (one launch)
  std::vector<lb_pack_ent> pack_tab;
  lb_pack_ent* pack_dev = nullptr;
  float* wpack = nullptr;
  int64_t wpack_floats = 0, wpack_cap = 0;
  // f16x2 arithmetic of the tall-skinny products (k_lin32h, round 5): fp16 hi / lo fragments of the operand matrices in the
  // same packed blob + one inverse power-of-two scale per matrix.  LB_TRAIN_MATH=f32 keeps the exact-fp32 kernels.
  bool f16x2 = lb_train_f16x2_default();
  // Range of k_dw_part_h's X operand (saved activations; round 6).  dY is scaled per row chunk inside the
  // kernel; X is multiplied by ONE power of two per call site (operand of the step: the c-th k_dw_part_h launch of every
  // step has the same producer), 2^dw_xexp[c], initially 1, under a range guard: a block whose largest scaled |X| leaves
  // [2^-8, 2^15) raises dw_flag[0] and leaves the largest raw |X| of the call in dw_flag[1 + c].  The step's gradient
  // reductions then add NOTHING, lb_gns_train_loss_grad re-centres the exponents of the calls that fired and repeats the
  // step.  A call whose chunks do not fit one window (it fires again right after a re-centring) switches to per-chunk
  // scaling inside the kernel (dw_xdyn[c]: one more pass over its X slice).
  // MLP depth (round 6: any num_mlp_layers >= 2, models/utils.py:100-115): a block keeps nlin - 1 hidden activations, the
  // first at its `a` pointer, the next ones hs_n / hs_e floats further on (node- / edge-sized blocks)
  int nlin = 2;
  int64_t hs_n = 0, hs_e = 0;
  float* da2 = nullptr;         // second d(hidden) buffer: the middle Linears' backward ping-pongs between da and da2
  int32_t* dw_flag = nullptr;   // [1 + LB_DW_CALLS] (bit 4 of [0]: k_sender_transpose met an edge without its transpose)
  int32_t dw_fallbacks = 0;     // steps repeated because of the guard
  int dw_call = 0;              // k_dw_part_h launches of the current step so far
  std::vector<int8_t> dw_xexp;  // per call site
  std::vector<uint8_t> dw_xdyn;
  bool cub_sort = false;        // this step's sender-sorted view comes from the radix sort
  float* tmax = nullptr;       // [rows / 16] largest |X| per row tile of the last k_lin32h call that was asked for it
  bool tmax_ok = false;        // ... and whether that call ran on k_lin32h
  std::vector<lb_pack_ent_h> pack_tab_h;
  lb_pack_ent_h* pack_dev_h = nullptr;
  float* wsc = nullptr;
};
#define LB_PACK_MAX 1024

// ---------------------------------------------------------------------------------------------- kernels
// column sums of x (rows x cols, cols <= 128): deterministic two-level reduction (fixed row blocks of 128 rows)
__global__ void k_colsum_part(const float* __restrict__ x, int64_t rows, int cols, int ld, float* __restrict__ part) {
  const int c = threadIdx.x;
  if (c >= cols) return;
  const int64_t r0 = (int64_t)blockIdx.x * 128, r1 = r0 + 128 < rows ? r0 + 128 : rows;
  float s = 0.f;
  for (int64_t r = r0; r < r1; ++r) s += x[r * ld + c];
  part[(int64_t)blockIdx.x * 128 + c] = s;
}
// ---- weight gradient (round 4): dW[K x 128] += X^T dY and db[128] += column sums of dY, without the library.
// rocBLAS ran this contraction - reduction over ~1e5 rows into a 384 x 128 result - at 16 TFLOP/s (660 us per call, 29 %
// of a TGV3D training step) and the bias sums went through a 128-thread serial loop (another 50 %).  Split over the ROWS:
// workgroup g owns a contiguous chunk of rows and forms its partial X^T dY on v_mfma_f32_16x16x4_f32 (exact fp32 products,
// fp32 accumulate); a reduction step is 4 rows (lane (i, kk) works on row r0 + kk).  Wave w = (mh, kq): column half mh
// (64 columns of dY) x quarter kq of K (32 NA columns of X, NA = ceil(K / 128)).  Operands come straight from global memory
// with WIDE loads and the MFMA index maps are permuted to fit them: a lane's one 16-byte load of dY[r][64 mh + 4 i .. + 3]
// is the B operand of FOUR column tiles (tile q holds the columns 64 mh + 4 i + q), its 8-byte load of
// X[r][c0 + 32 a + 2 i .. + 1] the A operand of TWO row tiles (tile (a, j): rows c0 + 32 a + 2 i + j) - 1 + NA loads per
// 8 NA MFMAs (a first version with one dword per operand and MFMA ran at 61 TFLOP/s, load-issue bound).  The column sums
// of dY are the running sums of the B values.  Partials go to part[g][K + 1][128] and are summed over g in ascending order
// by k_part_reduce: bit-reproducible, no atomics.
template <int NA, int DEPTH>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) k_dw_part(const float* __restrict__ X, int ldx, int K, const float* __restrict__ dY,
                                                  int64_t rows, int64_t chunk, float* __restrict__ part) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int i = lane & 15, kk = lane >> 4;
  const int mh = w & 1, kq = w >> 1;
  const int c0 = kq * 32 * NA;
  const int64_t r_begin = (int64_t)blockIdx.x * chunk, r_end = r_begin + chunk < rows ? r_begin + chunk : rows;
  f32x4 acc[NA][2][4];
#pragma unroll
  for (int a = 0; a < NA; ++a)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[a][j][q] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 bs = {0.f, 0.f, 0.f, 0.f};
  bool kok[NA];
#pragma unroll
  for (int a = 0; a < NA; ++a) kok[a] = c0 + 32 * a + 2 * i < K;  // (K is even: 8, a multiple of 32, ...)
  // software pipeline: the operands of DEPTH reduction steps are in flight (the compiler does not hoist the loads of an
  // unrolled loop above the MFMAs of the previous steps; gfx9 returns loads in order, so consuming the oldest slot waits
  // for exactly that slot)
  // DEPTH reduction steps in flight.  Round 5: 12 (NA 1) / 8 (NA 2) for long chunks - 8 MFMAs per step and wave do not cover the
  // HBM latency with 6 (TGV3D step -1.5 %, SEGNN -2.5 %); short chunks (node-sized products) keep 6: a deeper prologue fetches
  // past the chunk's end (TGV2D step +4 % with 12 everywhere)
  f32x4 bq[DEPTH];
  f32x2 aq[DEPTH][NA];
  auto fetch = [&](int64_t r0, f32x4& b, f32x2 (&av)[NA]) {
    const int64_t r = r0 + kk;
    const bool ok = r < r_end;
    b = ok ? *reinterpret_cast<const f32x4*>(dY + r * 128 + 64 * mh + 4 * i) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int a = 0; a < NA; ++a)
      av[a] = (ok && kok[a]) ? *reinterpret_cast<const f32x2*>(X + r * ldx + c0 + 32 * a + 2 * i) : f32x2{0.f, 0.f};
  };
  if (c0 < K || kq == 0) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) fetch(r_begin + 4 * d, bq[d], aq[d]);
    for (int64_t r0 = r_begin; r0 < r_end; r0 += 4 * DEPTH) {
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) {
        const f32x4 b = bq[d];
        f32x2 av[NA];
#pragma unroll
        for (int a = 0; a < NA; ++a) av[a] = aq[d][a];
        fetch(r0 + 4 * (d + DEPTH), bq[d], aq[d]);
        bs = bs + b;
#pragma unroll
        for (int a = 0; a < NA; ++a)
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[a][j][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[a][j], b[q], acc[a][j][q], 0, 0, 0);
      }
    }
  }
  // D layout: element jj of the tile of lane (i, kk) is (tile row 4 kk + jj, tile column i):
  //   dW row = c0 + 32 a + 2 (4 kk + jj) + j, column = 64 mh + 4 i + q
  float* out = part + (int64_t)blockIdx.x * (K + 1) * 128;
#pragma unroll
  for (int a = 0; a < NA; ++a)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const int row = c0 + 32 * a + 2 * (4 * kk + jj) + j;
        if (row < K)
          *reinterpret_cast<f32x4*>(out + row * 128 + 64 * mh + 4 * i) =
              f32x4{acc[a][j][0][jj], acc[a][j][1][jj], acc[a][j][2][jj], acc[a][j][3][jj]};
      }
  if (kq == 0) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float v = bs[q];
      v += __shfl_xor(v, 16);
      v += __shfl_xor(v, 32);
      bs[q] = v;
    }
    if (kk == 0) *reinterpret_cast<f32x4*>(out + K * 128 + 64 * mh + 4 * i) = bs;
  }
}
// ---- the same contraction in f16x2 arithmetic (round 5, second half; K >= 32 in slices of 128 X columns: every processor /
// decoder Linear of GNS, SEGNN's stacked operands).
// dW = X^T dY reduces over the ROWS, so the fp16 operands of v_mfma_f32_16x16x32_f16 hold eight consecutive rows of ONE column:
// a transpose of the row-major operands.  It goes through LDS, fused with the hi / lo split: thread (g, c) of the workgroup
// loads the eight rows 8 g .. 8 g + 7 of column c of X and of dY of a 32-row step (each load instruction of a wave is 256
// contiguous bytes of one row), splits them once - no wave repeats another's conversion - and stores them as two 16-byte
// fragments per matrix, exactly the registers the MFMA wants: entry (tile * 2 + part) * 64 + (c & 15) + 16 g.  Wave (ah, bq)
// multiplies the four tiles of X-column half ah with the two tiles of dY-column quarter bq: 24 MFMAs per step and wave,
// 12 16-byte LDS reads; LDS double-buffered, one barrier per step; the next step's global loads are in flight meanwhile.
// Range: X (activations) is split as it is; dY (gradients of any size) is multiplied by the power of two that puts the
// largest |dY| of THIS workgroup's row chunk into [1, 2) - found in a first pass over the chunk (it is read again from L2 /
// Infinity Cache) - and the partial result by its inverse: error <= 2^-22 of (|x| x chunk maximum of |dY|) per term, the
// bound a global scale gives; rows far below their chunk's maximum contribute with less relative precision and
// correspondingly little weight.  grid (G, K / 128); block (g, a) forms rows 128 a .. 128 a + 127 of part[g] and, for a = 0,
// the column sums of dY (exact fp32 sums of the unscaled values).
__global__ void __launch_bounds__(512) k_dw_part_h(const float* __restrict__ X, int ldx, int K, const float* __restrict__ dY,
                                                   int64_t rows, int64_t chunk, float* __restrict__ part,
                                                   const float* __restrict__ tmax, int32_t* __restrict__ xflag,
                                                   int32_t* __restrict__ xslot, int xexp, int xscale,
                                                   const float* __restrict__ dY_b, float* __restrict__ part_b) {
  __shared__ h8 sAB[2][2][1024];   // [buffer][A | B][(tile * 2 + part) * 64 + lane]: 64 KiB
  // blockIdx.z = 1: the second product of a paired launch - the same X against another dY (dw_acc_pair: the sender / receiver
  // halves of the edge block's first Linear share the node latents; two node-sized launches of ~6 us were two latency chains)
  if (blockIdx.z) {
    dY = dY_b;
    part = part_b;
  }
  float* red = reinterpret_cast<float*>(&sAB[0][0][0]);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = tid & 127, g = tid >> 7;
  const int ablk = blockIdx.y;
  // K need not be a multiple of 128 (SEGNN's stacked operands: 64, 144 ..): a thread whose X column lies past K reads the
  // row's last column instead and multiplies by zero when the value is used
  const int cx = 128 * ablk + c < K ? c : K - 1 - 128 * ablk;
  const float xkeep = 128 * ablk + c < K ? 1.f : 0.f;
  X += 128 * ablk;
  const int64_t r_begin = (int64_t)blockIdx.x * chunk, r_end = r_begin + chunk < rows ? r_begin + chunk : rows;
  // pass 0: the chunk's largest |dY|
  float m = 0.f;
  if (tmax) {   // the producer's row-tile maxima (k_lin32h read dY as its X operand just before): the tiles this chunk touches
    for (int64_t i = (r_begin >> 4) + tid; i < ((r_end + 15) >> 4); i += 512) m = fmaxf(m, tmax[i]);
  } else {
    const f32x4* y4 = reinterpret_cast<const f32x4*>(dY + r_begin * 128);
    const int64_t n4 = (r_end - r_begin) * 32;
    int64_t i = tid;
    for (; i + 7 * 512 < n4; i += 8 * 512) {   // eight loads in flight per thread
      f32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = y4[i + u * 512];
#pragma unroll
      for (int u = 0; u < 8; ++u) m = fmaxf(m, fmaxf(fmaxf(fabsf(v[u][0]), fabsf(v[u][1])), fmaxf(fabsf(v[u][2]), fabsf(v[u][3]))));
    }
    for (; i < n4; i += 512) {
      const f32x4 v = y4[i];
      m = fmaxf(m, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
    }
  }
  red[tid] = m;
  __syncthreads();
  for (int o = 256; o > 0; o >>= 1) {
    if (tid < o) red[tid] = fmaxf(red[tid], red[tid + o]);
    __syncthreads();
  }
  m = red[0];
  __syncthreads();
  unsigned ex = (__float_as_uint(m) >> 23) & 0xffu;
  ex = m == 0.f ? 127u : (ex < 1u ? 1u : (ex > 253u ? 253u : ex));
  const float sc = __uint_as_float((254u - ex) << 23), inv = __uint_as_float(ex << 23);
  // X (activations): multiplied by the call site's power of two 2^xexp (host side, lb_gns_train::dw_xexp) under the range
  // guard at the end of the kernel; xscale (a call site whose chunks do not fit one window): dY's treatment - a first pass over
  // this block's 128-column slice of the chunk finds the largest |X| (the slice is read again right after), the operand is
  // multiplied by the power of two that puts it into [1, 2) and the partial result by its inverse.
  float scx = __uint_as_float((unsigned)(127 + xexp) << 23), invx = __uint_as_float((unsigned)(127 - xexp) << 23);
  if (xscale) {
    float mx = 0.f;
    const float* xc = X + cx;
    int64_t r = r_begin + g;
    for (; r + 28 < r_end; r += 32) {   // eight rows in flight per thread (each load of a wave: 256 contiguous bytes of a row)
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = xc[(r + 4 * u) * ldx];
#pragma unroll
      for (int u = 0; u < 8; u += 2) mx = fmaxf(fmaxf(fabsf(v[u]), fabsf(v[u + 1])), mx);
    }
    for (; r < r_end; r += 4) mx = fmaxf(mx, fabsf(xc[r * ldx]));
    mx *= xkeep;
    red[tid] = mx;
    __syncthreads();
    for (int o = 256; o > 0; o >>= 1) {
      if (tid < o) red[tid] = fmaxf(red[tid], red[tid + o]);
      __syncthreads();
    }
    mx = red[0];
    __syncthreads();
    unsigned exx = (__float_as_uint(mx) >> 23) & 0xffu;
    exx = mx == 0.f ? 127u : (exx < 1u ? 1u : (exx > 253u ? 253u : exx));
    scx = __uint_as_float((254u - exx) << 23);
    invx = __uint_as_float(exx << 23);
  }

  // three steps' operands in flight (16 dwords per step and thread): one step of compute does not cover the HBM latency.
  // Addressing costs no VALU: a load is (uniform base of the step) + (this thread's row offset, computed once); the main loop
  // runs over the FULL 32-row steps only - no clamp, no mask, no branch inside its body (a select behind a load waits for
  // the load, a branch inside the unrolled body makes the compiler wait for every outstanding load: both were measured) -
  // look-ahead fetches past the last full step re-read that step (unused); the chunk's last n % 32 rows are one masked step
  // after the loop.
  const int64_t n = r_end - r_begin;
  const int full = (int)(n >> 5);
  const float* Xc = X + r_begin * ldx + cx;
  const float* Yc = dY + r_begin * 128 + c;
  int xo[8], yo[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    xo[t] = (8 * g + t) * ldx;
    yo[t] = (8 * g + t) * 128;
  }
  float xq[3][8], yq[3][8];
  auto fetch = [&](int step, float (&xv)[8], float (&yv)[8]) {
    const int st = step < full ? step : (full > 0 ? full - 1 : 0);   // (uniform)
    const float* xb = Xc + (int64_t)st * 32 * ldx;
    const float* yb = Yc + (int64_t)st * 32 * 128;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      xv[t] = xb[xo[t]];
      yv[t] = yb[yo[t]];
    }
  };
  const int ah = wave & 1, bq = wave >> 1;
  f32x4 acc[4][2];
#pragma unroll
  for (int ta = 0; ta < 4; ++ta)
#pragma unroll
    for (int tb = 0; tb < 2; ++tb) acc[ta][tb] = f32x4{0.f, 0.f, 0.f, 0.f};
  float bsum = 0.f;
  float xmax = 0.f;   // largest |X| this thread split (v_max3_f32 with |abs| modifiers: four instructions per step)
  const float xk = xkeep * scx;
  const int wslot = ((c >> 4) * 2) * 64 + (c & 15) + 16 * g;
  int buf = 0;
  auto mma_step = [&](const f32x4& x0, const f32x4& x1, const f32x4& y0, const f32x4& y1) {
    bsum += (((y0[0] + y0[1]) + (y0[2] + y0[3])) + ((y1[0] + y1[1]) + (y1[2] + y1[3])));
    xmax = fmaxf(fmaxf(fabsf(x0[0]), fabsf(x0[1])), xmax);
    xmax = fmaxf(fmaxf(fabsf(x0[2]), fabsf(x0[3])), xmax);
    xmax = fmaxf(fmaxf(fabsf(x1[0]), fabsf(x1[1])), xmax);
    xmax = fmaxf(fmaxf(fabsf(x1[2]), fabsf(x1[3])), xmax);
    h8 xh, xl, yh, yl;
    lb_split8v(x0 * xk, x1 * xk, xh, xl);
    lb_split8v(y0 * sc, y1 * sc, yh, yl);
    sAB[buf][0][wslot] = xh;
    sAB[buf][0][wslot + 64] = xl;
    sAB[buf][1][wslot] = yh;
    sAB[buf][1][wslot + 64] = yl;
    __syncthreads();
    h8 bh[2], bl[2];
#pragma unroll
    for (int tb = 0; tb < 2; ++tb) {
      bh[tb] = sAB[buf][1][((2 * bq + tb) * 2) * 64 + lane];
      bl[tb] = sAB[buf][1][((2 * bq + tb) * 2 + 1) * 64 + lane];
    }
#pragma unroll
    for (int ta = 0; ta < 4; ++ta) {
      const h8 ahh = sAB[buf][0][((4 * ah + ta) * 2) * 64 + lane], all_ = sAB[buf][0][((4 * ah + ta) * 2 + 1) * 64 + lane];
#pragma unroll
      for (int tb = 0; tb < 2; ++tb) {
        acc[ta][tb] = MFMA16H(ahh, bh[tb], acc[ta][tb]);
        acc[ta][tb] = MFMA16H(ahh, bl[tb], acc[ta][tb]);
        acc[ta][tb] = MFMA16H(all_, bh[tb], acc[ta][tb]);
      }
    }
    buf ^= 1;
  };
  if (full > 0) {
#pragma unroll
    for (int d = 0; d < 3; ++d) fetch(d, xq[d], yq[d]);
    int st = 0;
    for (; st + 3 <= full; st += 3) {
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        const f32x4 x0 = {xq[d][0], xq[d][1], xq[d][2], xq[d][3]}, x1 = {xq[d][4], xq[d][5], xq[d][6], xq[d][7]};
        const f32x4 y0 = {yq[d][0], yq[d][1], yq[d][2], yq[d][3]}, y1 = {yq[d][4], yq[d][5], yq[d][6], yq[d][7]};
        __builtin_amdgcn_sched_barrier(0);   // (the compiler otherwise sinks these loads to their use, three steps later)
        fetch(st + d + 3, xq[d], yq[d]);
        __builtin_amdgcn_sched_barrier(0);
        mma_step(x0, x1, y0, y1);
      }
    }
    // the one or two full steps left over: their operands are in stages 0 (and 1)
    if (st < full) {
      const f32x4 x0 = {xq[0][0], xq[0][1], xq[0][2], xq[0][3]}, x1 = {xq[0][4], xq[0][5], xq[0][6], xq[0][7]};
      const f32x4 y0 = {yq[0][0], yq[0][1], yq[0][2], yq[0][3]}, y1 = {yq[0][4], yq[0][5], yq[0][6], yq[0][7]};
      mma_step(x0, x1, y0, y1);
    }
    if (st + 1 < full) {
      const f32x4 x0 = {xq[1][0], xq[1][1], xq[1][2], xq[1][3]}, x1 = {xq[1][4], xq[1][5], xq[1][6], xq[1][7]};
      const f32x4 y0 = {yq[1][0], yq[1][1], yq[1][2], yq[1][3]}, y1 = {yq[1][4], yq[1][5], yq[1][6], yq[1][7]};
      mma_step(x0, x1, y0, y1);
    }
  }
  if (n & 31) {   // the chunk's last rows: clamped loads, zeros past the end
    const int64_t rb = r_begin + (int64_t)full * 32 + 8 * g;
    f32x4 x0, x1, y0, y1;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int64_t r = rb + t;
      const bool ok = r < r_end;
      const int64_t rc = ok ? r : r_end - 1;
      const float xv = ok ? X[rc * ldx + cx] : 0.f, yv = ok ? dY[rc * 128 + c] : 0.f;
      if (t < 4) x0[t] = xv, y0[t] = yv;
      else x1[t - 4] = xv, y1[t - 4] = yv;
    }
    mma_step(x0, x1, y0, y1);
  }
  // D[i][j] of tile (ta, tb): lane (j = lane & 15, 4 (lane >> 4) + v = i): X column 16 (4 ah + ta) + i, dY column 16 (2 bq + tb) + j
  float* out = part + (int64_t)blockIdx.x * (K + 1) * 128 + (int64_t)128 * ablk * 128;
#pragma unroll
  for (int ta = 0; ta < 4; ++ta)
#pragma unroll
    for (int tb = 0; tb < 2; ++tb)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int xr = 16 * (4 * ah + ta) + 4 * (lane >> 4) + v;
        if (128 * ablk + xr < K) out[xr * 128 + 16 * (2 * bq + tb) + (lane & 15)] = (acc[ta][tb][v] * inv) * invx;
      }
  // X range guard: the chunk's largest |X| (this block's 128-column slice).  >= 2^15: the hi half is about to leave fp16
  // (65504; infinities land here too); < 2^-8 and not all zero: every lo half of the slice is a fp16 subnormal, the terms
  // keep 2^-25 absolute instead of 2^-22 relative precision.  Either way the whole step is redone in exact fp32.
  {
    float m = xmax * xkeep;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    __syncthreads();   // (the last step's fragments have been read)
    if (lane == 0) red[512 + wave] = m;
    __syncthreads();
    if (tid == 0) {
      float mm = red[512];
#pragma unroll
      for (int w = 1; w < 8; ++w) mm = fmaxf(mm, red[512 + w]);
      const float ms = mm * scx;   // (mm: raw values)
      if (xflag && !xscale && (!(ms < 32768.f) || (ms > 0.f && ms < 0.00390625f))) {
        atomicOr(xflag, ms < 1.f ? 2 : 1);
        atomicMax(reinterpret_cast<unsigned*>(xslot), __float_as_uint(mm));   // (non-negative floats order like their bits)
      }
    }
  }
  if (ablk == 0) {
    red[tid] = bsum;
    __syncthreads();
    if (tid < 128)
      part[(int64_t)blockIdx.x * (K + 1) * 128 + (int64_t)K * 128 + tid] = ((red[tid] + red[128 + tid]) + red[256 + tid]) + red[384 + tid];
  }
}
// Ordered sum of partials: out e < n0: dst0[e] += sum_g part[g * stride + e]; n0 <= e < n0 + n1: dst1[e - n0] += sum_g
// part[g * stride + off1 + e - n0].  A 1024-thread block owns 256 outputs x 16 ranges of g: a lane owns four consecutive outputs
// (one 16-byte load per partial, four in flight; the round-5 kernel read 4 bytes per lane and partial - 256-byte
// requests 66 KB apart, 2.1 TB/s on 350 MB of partials = 165 us of a 2.4 ms TGV2D step), each added in ascending g, the
// 16 range sums are combined in range order through LDS: the result does not depend on timing (and equals the round-5
// kernel's bit for bit: the order per output is the same).  A quad that straddles n0 / the end, or whose source is not
// 16-byte aligned (k_dw_narrow's K x M slots), takes the same sums with scalar loads.
// Round 5: ONE launch per training step for all reductions (88 of them: every weight / bias / LayerNorm gradient) - each
// producer writes its partials into a slot of its own and leaves a descriptor; a flat grid, each block finds its descriptor by bisection.  (One
// launch per producer before: 0.5 ms of a 7 ms TGV3D step, 0.4 ms of a 3.5 ms TGV2D step, mostly launch latency.)
#define RED_OUT 256   // outputs per block of k_part_reduce (red_push counts its blocks with the same number)
__global__ void __launch_bounds__(1024) k_part_reduce(const float* __restrict__ part_base, const lb_red_ent* __restrict__ tab,
                                                      int n_ent, float* __restrict__ grad, const int32_t* __restrict__ skip) {
  __shared__ f32x4 s_red[16][64];
  if (skip && *skip) return;   // k_dw_part_h's range guard fired: this step adds nothing, the host repeats it in fp32
  int lo = 0, hi = n_ent - 1;  // the last descriptor with blk0 <= blockIdx.x
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (tab[mid].blk0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const lb_red_ent d = tab[lo];
  const int bx = (int)blockIdx.x - d.blk0;
  const float* part = part_base + d.part;
  const int G = d.G, n0 = d.n0, n_all = d.n0 + d.n1;
  const int64_t stride = d.stride;
  const int c = threadIdx.x & 63, seg = threadIdx.x >> 6;
  const int e0 = bx * RED_OUT + 4 * c;
  const int per = (G + 15) / 16, g0 = seg * per, g1 = g0 + per < G ? g0 + per : G;
  // source of output e0 + j; a quad is "whole" when its four outputs are consecutive floats of one region
  const bool in0 = e0 + 3 < n0, in1 = e0 >= n0 && e0 + 3 < n_all;
  const int64_t src0 = e0 < n0 ? e0 : (int64_t)d.off1 + (e0 - n0);
  const bool vec = (in0 || in1) && ((src0 | stride | d.part) & 3) == 0;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (vec) {
    const float* p = part + src0;
    int g = g0;
    for (; g + 4 <= g1; g += 4) {
      const f32x4 v0 = *reinterpret_cast<const f32x4*>(p + (int64_t)g * stride);
      const f32x4 v1 = *reinterpret_cast<const f32x4*>(p + (int64_t)(g + 1) * stride);
      const f32x4 v2 = *reinterpret_cast<const f32x4*>(p + (int64_t)(g + 2) * stride);
      const f32x4 v3 = *reinterpret_cast<const f32x4*>(p + (int64_t)(g + 3) * stride);
      s = (((s + v0) + v1) + v2) + v3;
    }
    for (; g < g1; ++g) s = s + *reinterpret_cast<const f32x4*>(p + (int64_t)g * stride);
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int e = e0 + j;
      if (e >= n_all) continue;
      const int64_t src = e < n0 ? e : (int64_t)d.off1 + (e - n0);
      float sj = 0.f;
      int g = g0;
      for (; g + 4 <= g1; g += 4) {
        const float v0 = part[(int64_t)g * stride + src], v1 = part[(int64_t)(g + 1) * stride + src];
        const float v2 = part[(int64_t)(g + 2) * stride + src], v3 = part[(int64_t)(g + 3) * stride + src];
        sj = (((sj + v0) + v1) + v2) + v3;
      }
      for (; g < g1; ++g) sj += part[(int64_t)g * stride + src];
      s[j] = sj;
    }
  }
  s_red[seg][c] = s;
  __syncthreads();
  if (seg == 0 && e0 < n_all) {
    f32x4 v = s_red[0][c];
#pragma unroll
    for (int k = 1; k < 16; ++k) v = v + s_red[k][c];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int e = e0 + j;
      if (e < n0) grad[d.dst0 + e] += v[j];
      else if (e < n_all) grad[d.dst1 + (e - n0)] += v[j];
    }
  }
}
// ---- tall-skinny products without the library (round 5): k_pack_w / k_lin32 / k_lin32f of lb_lin32.h

// dW[K x M] += X^T dY for a NARROW dY (M <= 4: the decoder's output Linear), which k_dw_part's 128-column tiling does not
// cover: thread c of a 128-thread workgroup owns column c of X over a contiguous chunk of rows; partials
// part[g][K][4] are summed over g in ascending order by k_part_reduce.
__global__ void __launch_bounds__(128) k_dw_narrow(const float* __restrict__ X, int ldx, int K, const float* __restrict__ dY,
                                                    int ldy, int M, int64_t rows, int64_t chunk, float* __restrict__ part) {
  const int c = threadIdx.x;
  const int64_t r0 = (int64_t)blockIdx.x * chunk, r1 = r0 + chunk < rows ? r0 + chunk : rows;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  if (c < K)
    for (int64_t r = r0; r < r1; ++r) {
      const float x = X[r * ldx + c];
#pragma unroll
      for (int m = 0; m < 4; ++m)
        if (m < M) acc[m] += x * dY[r * ldy + m];
    }
  if (c < K)
#pragma unroll
    for (int m = 0; m < 4; ++m)
      if (m < M) part[((int64_t)blockIdx.x * K + c) * M + m] = acc[m];
}

// LayerNorm backward with the parameter gradients folded in (round 4; k_ln_bwd + two column-sum passes over a scratch copy
// of dy * zhat before): a workgroup of 4 waves walks LNB_ROWS rows, every lane keeps the running sums of dy * zhat and dy of
// its two columns, the four waves are combined through LDS in wave order -> part[block][2][128] (k_part_reduce sums the blocks
// in ascending order).
#define LNB_ROWS 64
// Round 5, second half: a row is held by SIXTEEN lanes (eight consecutive columns each: two 16-byte loads per array instead of
// two dwords), four rows per wave and step, and the row sums are four DPP steps inside the 16-lane row (quad_perm x 2,
// row_half_mirror, row_mirror: no LDS crossbar) instead of six dependent ds_bpermute round trips over 64 lanes - the first
// version spent 48 us on 170 MB per edge-sized call.  The per-column sums of dy * zhat and dy stay in the lane that owns the
// column across its rows and are combined over the four row slots and four waves in a fixed order.
__device__ __forceinline__ float lb_row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));   // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));  // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, true));  // row_mirror
  return v;
}
template <bool GATHER>
__global__ void __launch_bounds__(256) k_ln_bwd2(const float* __restrict__ z, const float* __restrict__ b1,
                                                 const float* __restrict__ sc, const float* __restrict__ dy,
                                                 float* __restrict__ dz, int64_t rows, float* __restrict__ part, int d,
                                                 const float* __restrict__ gth, const int32_t* __restrict__ gidx) {
  // gth != null: the incoming gradient is dy[r] + gth[gidx[r]] - the transpose of jraph.segment_sum (the aggregate's gradient
  // gathered over the receivers) folded into this load (round 4: k_seg_sum_bwd, a pass of its own over E x 128)
  __shared__ float s_red[4][4][2][128];   // [wave][row slot][dy * zhat | dy][column]
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int cl = lane & 15, rq = lane >> 4;     // columns 8 cl .. 8 cl + 7 of row slot rq
  const f32x4 scA = reinterpret_cast<const f32x4*>(sc)[2 * cl], scB = reinterpret_cast<const f32x4*>(sc)[2 * cl + 1];
  const f32x4 bA = reinterpret_cast<const f32x4*>(b1)[2 * cl], bB = reinterpret_cast<const f32x4*>(b1)[2 * cl + 1];
  f32x4 psA = {0.f, 0.f, 0.f, 0.f}, psB = psA, poA = psA, poB = psA;
  const float inv_d = 1.f / (float)d;
  f32x4 keepA, keepB;   // 1 in the real columns, 0 in the padding of a latent narrower than 128
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    keepA[j] = 8 * cl + j < d ? 1.f : 0.f;
    keepB[j] = 8 * cl + 4 + j < d ? 1.f : 0.f;
  }
  const int64_t rb = (int64_t)blockIdx.x * LNB_ROWS + 16 * wv;   // this wave's 16 rows: four steps of four
  constexpr int NIT = 4;
  f32x4 zA[NIT], zB[NIT], gA[NIT], gB[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {   // every load of the wave's rows before the first reduction
    const int64_t r = rb + 4 * it + rq;
    const int64_t rc = r < rows ? r : rows - 1;
    const f32x4* zr = reinterpret_cast<const f32x4*>(z + rc * TD) + 2 * cl;
    const f32x4* gr = reinterpret_cast<const f32x4*>(dy + rc * TD) + 2 * cl;
    zA[it] = zr[0]; zB[it] = zr[1];
    gA[it] = gr[0]; gB[it] = gr[1];
  }
  if (GATHER) {  // (a compile-time branch: a run-time one inside the loop above serialises its loads)
    int gi[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int64_t r = rb + 4 * it + rq;
      gi[it] = gidx[r < rows ? r : rows - 1];
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const f32x4* tr = reinterpret_cast<const f32x4*>(gth + (int64_t)gi[it] * TD) + 2 * cl;
      gA[it] = tr[0] + gA[it];
      gB[it] = tr[1] + gB[it];
    }
  }
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int64_t r0 = rb + 4 * it + rq;
    const bool ok = r0 < rows;
    const int64_t r = ok ? r0 : rows - 1;   // (a row past the end recomputes the last row and stores the same values again)
    const f32x4 xA = zA[it] + bA, xB = zB[it] + bB;
    const float mean = lb_row16_sum(((xA[0] + xA[1]) + (xA[2] + xA[3])) + ((xB[0] + xB[1]) + (xB[2] + xB[3]))) * inv_d;
    f32x4 dA, dB;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      dA[j] = xA[j] - mean;
      dB[j] = xB[j] - mean;
    }
    float q = lb_row16_sum(((dA[0] * dA[0] + dA[1] * dA[1]) + (dA[2] * dA[2] + dA[3] * dA[3])) +
                           ((dB[0] * dB[0] + dB[1] * dB[1]) + (dB[2] * dB[2] + dB[3] * dB[3])));
    q -= (float)(TD - d) * mean * mean;   // the padded columns hold 0 - mean
    const float rs = 1.0f / sqrtf(q * inv_d + 1e-5f);
    const f32x4 hA = dA * rs, hB = dB * rs;
    const f32x4 uA = gA[it] * scA, uB = gB[it] * scB;
    float a = ((uA[0] + uA[1]) + (uA[2] + uA[3])) + ((uB[0] + uB[1]) + (uB[2] + uB[3]));
    float b = ((uA[0] * hA[0] + uA[1] * hA[1]) + (uA[2] * hA[2] + uA[3] * hA[3])) +
              ((uB[0] * hB[0] + uB[1] * hB[1]) + (uB[2] * hB[2] + uB[3] * hB[3]));
    a = lb_row16_sum(a) * inv_d;
    b = lb_row16_sum(b) * inv_d;
    f32x4 oA, oB;
#pragma unroll
    for (int j = 0; j < 4; ++j) {   // (u is 0 in the padded columns: they add nothing to a, b)
      oA[j] = keepA[j] * (rs * (uA[j] - a - hA[j] * b));
      oB[j] = keepB[j] * (rs * (uB[j] - a - hB[j] * b));
    }
    f32x4* dr = reinterpret_cast<f32x4*>(dz + r * TD) + 2 * cl;
    dr[0] = oA;
    dr[1] = oB;
    const float w = ok ? 1.f : 0.f;
    psA = psA + gA[it] * hA * w; psB = psB + gB[it] * hB * w;
    poA = poA + gA[it] * w;      poB = poB + gB[it] * w;
  }
  f32x4* sr = reinterpret_cast<f32x4*>(&s_red[wv][rq][0][0]) + 2 * cl;
  sr[0] = psA; sr[1] = psB;
  sr[32] = poA; sr[33] = poB;
  __syncthreads();
  const int c = threadIdx.x;  // 0..127 scale columns, 128..255 offset columns
  const int which = c >> 7, col = c & 127;
  float v = 0.f;
#pragma unroll
  for (int w2 = 0; w2 < 4; ++w2)
#pragma unroll
    for (int q2 = 0; q2 < 4; ++q2) v += s_red[w2][q2][which][col];
  part[(int64_t)blockIdx.x * 256 + c] = v;
}
// Edge block, first Linear, without the concatenation (round 4, as the inference kernels do it): W0 = [W_s ; W_r ; W_e] by
// rows, so  [n_s | n_r | e] W0 = (n W_s)[snd] + (n W_r)[rcv] + e W_e  - the two node-sized products are formed once per NODE
// (N rows) instead of once per EDGE (E ~ 14 N rows): the edge block's first Linear costs a third of the flops, forward
// and backward, and the E x 384 concatenated inputs are never stored.
//   a[e] = relu(e W_e + Ps[snd[e]] + Pr[rcv[e]] + b0)        (Ps = n W_s, Pr = n W_r; the gathers, the bias and the ReLU are
//   the epilogue of the edge-sized product: k_lin32f<4>)
// the transpose of the two gathers: dPs[i] = sum over the edges i SENDS of da[e] (sender-sorted permutation, ascending edge index),
// dPr[i] = sum over the edges i RECEIVES (its CSR row).  One 32-lane group per node, no atomics.
__global__ void k_edge_dP(const float* __restrict__ da, const int32_t* __restrict__ snd_ptr, const int32_t* __restrict__ snd_perm,
                          const int32_t* __restrict__ row_ptr, float* __restrict__ dPs, float* __restrict__ dPr, int64_t N,
                          int64_t E) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * 32) return;
  const int64_t r = i / 32;
  const int q = (int)(i % 32);
  f32x4 as = {0.f, 0.f, 0.f, 0.f}, ar = {0.f, 0.f, 0.f, 0.f};
  const int s0 = snd_ptr[r], s1 = snd_ptr[r + 1];
  for (int j = s0; j < s1; ++j) as = as + reinterpret_cast<const f32x4*>(da)[(int64_t)snd_perm[j] * 32 + q];
  int k0 = row_ptr[r], k1 = row_ptr[r + 1];
  k0 = k0 < E ? k0 : (int)E;
  k1 = k1 < E ? k1 : (int)E;
  for (int k = k0; k < k1; ++k) ar = ar + reinterpret_cast<const f32x4*>(da)[(int64_t)k * 32 + q];
  reinterpret_cast<f32x4*>(dPs)[i] = as;
  reinterpret_cast<f32x4*>(dPr)[i] = ar;
}
__global__ void k_iota(int32_t* __restrict__ x, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x[i] = (int32_t)i;
}
// snd_ptr[s] = first position of key >= s in the sorted sender keys (lower bound), s = 0 .. N
__global__ void k_lower_bounds(const int32_t* __restrict__ keys, int64_t E, int64_t N, int32_t* __restrict__ ptr) {
  const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s > N) return;
  int64_t lo = 0, hi = E;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (keys[mid] < (int32_t)s) lo = mid + 1; else hi = mid;
  }
  ptr[s] = (int32_t)lo;
}
// jraph.segment_sum on the receiver-sorted CSR (rows are contiguous edge ranges): agg[r] = sum_e msg[e]
// xn != null (the GNS node block): the sum goes straight into the right half of the node MLP's input row [n | agg] and the
// node latent n into the left half - the concatenation (k_concat_node_in until round 5: a launch per layer) rides along
__global__ void k_seg_sum(const int32_t* __restrict__ row_ptr, const float* __restrict__ msg, float* __restrict__ agg,
                          int64_t N, int64_t E, const float* __restrict__ n = nullptr, float* __restrict__ xn = nullptr) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * 32) return;
  const int64_t r = i / 32;
  const int q = (int)(i % 32);
  int64_t k0 = row_ptr[r], k1 = row_ptr[r + 1];
  k0 = k0 < E ? k0 : E;
  k1 = k1 < E ? k1 : E;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  for (int64_t k = k0; k < k1; ++k) s = s + reinterpret_cast<const f32x4*>(msg)[k * 32 + q];
  if (xn) {
    reinterpret_cast<f32x4*>(xn)[r * 64 + q] = reinterpret_cast<const f32x4*>(n)[i];
    reinterpret_cast<f32x4*>(xn)[r * 64 + 32 + q] = s;
  } else {
    reinterpret_cast<f32x4*>(agg)[i] = s;
  }
}
// non-kinematic particle count per trajectory (utils.py:28-35) and the per-node loss weight 1 / count
__global__ void k_count_nonkin(const int32_t* __restrict__ ptype, int64_t BN, int N, int32_t* __restrict__ cnt) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool in = i < BN;
  const int pt = in ? ptype[i] : 1;
  const bool nk = in && !(pt == 1 || pt == 2 || pt == -1);
  const int b = (int)((in ? i : BN - 1) / N);
  const int b_first = __builtin_amdgcn_readfirstlane(b);
  if (__all(b == b_first)) {  // the usual case: one integer atomic per wave (integer sums do not depend on the order)
    const int n = __popcll(__ballot(nk));
    if ((threadIdx.x & 63) == 0 && n) atomicAdd(&cnt[b_first], n);
  } else if (nk) {
    atomicAdd(&cnt[b], 1);
  }
}
__global__ void k_node_weight(const int32_t* __restrict__ ptype, const int32_t* __restrict__ cnt, int64_t BN, int N,
                              float* __restrict__ w) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= BN) return;
  const int pt = ptype[i];
  const bool kin = pt == 1 || pt == 2 || pt == -1;
  const int c = cnt[i / N];
  w[i] = (kin || c <= 0) ? 0.f : 1.0f / (float)c;
}
// loss = (1/B) sum_i w_i * lw * sum_d (pred - target)^2 ; dpred = 2 lw w_i (pred - target)   (gradients SUMMED over b)
__global__ void k_mse_grad(const float* __restrict__ pred, const float* __restrict__ target,
                           const float* __restrict__ nw, int64_t BN, int dim, float lw, float inv_b,
                           float* __restrict__ dpred, double* __restrict__ loss_part) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  double l = 0.0;
  if (i < BN) {
    const float w = nw[i];
    for (int d = 0; d < dim; ++d) {
      const float diff = pred[i * dim + d] - target[i * dim + d];
      l += (double)(w * lw * diff * diff);
      dpred[i * dim + d] = 2.f * lw * w * diff;
    }
  }
  for (int o = 32; o > 0; o >>= 1) l += __shfl_xor(l, o);
  // (round 4: per-wave partials summed in index order by k_loss_finish - the reported loss is bit-reproducible too)
  if ((threadIdx.x & 63) == 0) loss_part[((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6] = l * (double)inv_b;
}
__global__ void k_loss_finish(const double* __restrict__ part, int64_t n, double* __restrict__ loss) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double s = 0.0;
  for (int64_t i = 0; i < n; ++i) s += part[i];
  *loss += s;
}
// d embed[type] += sum over nodes of that type of dxnode[:, col0 : col0 + emb].  Deterministic (round 4; round 3 used fp32
// atomics): one 1024-thread block per type, thread (c, slice) sums column c over the nodes r = slice, slice + S, .. of that
// type in index order, the S = 1024 / emb slice sums are added in slice order.
__global__ void __launch_bounds__(1024) k_embed_grad(const float* __restrict__ dx, int ld, int col0, int emb,
                                                     const int32_t* __restrict__ ptype, int ntypes, int64_t BN,
                                                     float* __restrict__ gembed, const int32_t* __restrict__ skip) {
  __shared__ float s_part[1024];
  if (skip && *skip) return;   // (see k_part_reduce)
  const int type = blockIdx.x, c = threadIdx.x % emb, slice = threadIdx.x / emb, S = 1024 / emb;
  float acc = 0.f;
  if (slice < S) {
    // branch-free and eight rows at a time: a load behind `if (pt == type)` waits for the type, and the loop was a chain of
    // 125 dependent round trips (73 us on 8 k nodes); adding +0.f for the other types leaves the sum's bits unchanged
    int64_t r = slice;
    for (; r + 7 * S < BN; r += 8 * S) {
      int pt[8];
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        pt[u] = ptype[r + u * S];
        v[u] = dx[(r + u * S) * ld + col0 + c];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int q = pt[u] < 0 ? pt[u] + ntypes : pt[u];  // hk.Embed wraps negative ids (padding type -1 -> row 8)
        acc += q == type ? v[u] : 0.f;
      }
    }
    for (; r < BN; r += S) {
      int pt = ptype[r];
      if (pt < 0) pt += ntypes;
      const float v = dx[r * ld + col0 + c];
      acc += pt == type ? v : 0.f;
    }
  }
  s_part[threadIdx.x] = acc;
  __syncthreads();
  if ((int)threadIdx.x < emb) {
    float t = 0.f;
    for (int k = 0; k < S; ++k) t += s_part[k * emb + threadIdx.x];
    gembed[(int64_t)type * emb + threadIdx.x] += t;
  }
}
// optax.adamw(lr, b1, b2, eps, weight_decay): m, v moments, bias correction, decoupled decay (trainer.py:189-193)
__global__ void k_adamw(float* __restrict__ w, const float* __restrict__ g, float* __restrict__ m,
                        float* __restrict__ v, int64_t n, float lr, float b1, float b2, float eps, float wd,
                        float c1, float c2) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float gi = g[i];
  const float mi = b1 * m[i] + (1.f - b1) * gi;
  const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
  m[i] = mi;
  v[i] = vi;
  const float mh = mi / c1, vh = vi / c2;
  w[i] -= lr * (mh / (sqrtf(vh) + eps) + wd * w[i]);
}

// ------------------------------------------------------------------------------------------- host helpers
#define GRID1(n) dim3((unsigned)(((n) + 255) / 256)), dim3(256)

// the packed copy of Wop (NR x NO) = W (trans 0, row stride ldw) or W^T (trans 1): registered (and packed) at its first use
static int pack_lookup(lb_gns_train* t, const float* W, int NR, int NO, int ldw, int trans, int nob, const float** out) {
  const int64_t src = W - t->w;
  if (src < 0 || src >= t->n_floats) return lb_fail(LB_ERR_STATE, "k_lin32 operand outside the weight blob");
  for (const lb_pack_ent& e : t->pack_tab)
    if (e.src == src && e.NR == NR && e.NO == NO && e.ldw == ldw && e.trans == trans) {
      *out = t->wpack + e.dst;
      return LB_OK;
    }
  if (!t->wpack) {
    t->wpack_cap = 2 * t->n_floats + ((int64_t)1 << 21);
    LB_TRY(lb_alloc(&t->wpack, (size_t)t->wpack_cap));
    LB_TRY(lb_alloc(&t->pack_dev, (size_t)LB_PACK_MAX));
  }
  lb_pack_ent e{};
  e.src = src; e.dst = t->wpack_floats; e.NR = NR; e.NO = NO; e.ldw = ldw; e.trans = trans;
  e.NJ = (NR + 15) / 16; e.NOB = nob;
  const int64_t n = (int64_t)e.NJ * e.NOB * 256;
  if (t->wpack_floats + n > t->wpack_cap || t->pack_tab.size() >= LB_PACK_MAX)
    return lb_fail(LB_ERR_STATE, "k_lin32: packed operand table is full");
  const size_t idx = t->pack_tab.size();
  t->pack_tab.push_back(e);
  t->wpack_floats += n;
  LB_HIP(hipMemcpyAsync(t->pack_dev + idx, &t->pack_tab[idx], sizeof(lb_pack_ent), hipMemcpyHostToDevice, t->eng->stream));
  LB_HIP(hipStreamSynchronize(t->eng->stream));  // (the vector may move; first step only)
  hipLaunchKernelGGL(k_pack_w, dim3(16, 1), dim3(256), 0, t->eng->stream, t->w, t->wpack, t->pack_dev + idx);
  *out = t->wpack + e.dst;
  return LB_OK;
}
// the same for k_lin32h: fp16 hi / lo fragments times the matrix' power-of-two scale (k_pack_wh)
static int pack_lookup_h(lb_gns_train* t, const float* W, int NR, int NO, int ldw, int trans, const float** out, const float** wsc) {
  const int64_t src = W - t->w;
  if (src < 0 || src >= t->n_floats) return lb_fail(LB_ERR_STATE, "k_lin32h operand outside the weight blob");
  for (const lb_pack_ent_h& e : t->pack_tab_h)
    if (e.src == src && e.NR == NR && e.NO == NO && e.ldw == ldw && e.trans == trans) {
      *out = t->wpack + e.dst;
      *wsc = t->wsc + e.sc;
      return LB_OK;
    }
  if (!t->wpack) {
    t->wpack_cap = 2 * t->n_floats + ((int64_t)1 << 21);
    LB_TRY(lb_alloc(&t->wpack, (size_t)t->wpack_cap));
    LB_TRY(lb_alloc(&t->pack_dev, (size_t)LB_PACK_MAX));
  }
  if (!t->pack_dev_h) {
    LB_TRY(lb_alloc(&t->pack_dev_h, (size_t)LB_PACK_MAX));
    LB_TRY(lb_alloc(&t->wsc, (size_t)LB_PACK_MAX));
  }
  lb_pack_ent_h e{};
  e.src = src; e.dst = t->wpack_floats; e.NR = NR; e.NO = NO; e.ldw = ldw; e.trans = trans;
  e.NP = NR / 32; e.NOB = 8; e.sc = (int64_t)t->pack_tab_h.size();
  const int64_t n = (int64_t)e.NP * e.NOB * 512;
  if (t->wpack_floats + n > t->wpack_cap || t->pack_tab_h.size() >= LB_PACK_MAX)
    return lb_fail(LB_ERR_STATE, "k_lin32h: packed operand table is full");
  const size_t idx = t->pack_tab_h.size();
  t->pack_tab_h.push_back(e);
  t->wpack_floats += n;
  LB_HIP(hipMemcpyAsync(t->pack_dev_h + idx, &t->pack_tab_h[idx], sizeof(lb_pack_ent_h), hipMemcpyHostToDevice, t->eng->stream));
  LB_HIP(hipStreamSynchronize(t->eng->stream));  // (the vector may move; first step only)
  hipLaunchKernelGGL(k_pack_wh, dim3(16, 1), dim3(256), 0, t->eng->stream, t->w, t->wpack, t->wsc, t->pack_dev_h + idx);
  *out = t->wpack + e.dst;
  *wsc = t->wsc + e.sc;
  return LB_OK;
}
// every registered operand, from the current weights (the optimiser / lb_gns_train_write changed them)
static void pack_all(lb_gns_train* t) {
  if (!t->pack_tab.empty())
    hipLaunchKernelGGL(k_pack_w, dim3(16, (unsigned)t->pack_tab.size()), dim3(256), 0, t->eng->stream, t->w, t->wpack, t->pack_dev);
  if (!t->pack_tab_h.empty())
    hipLaunchKernelGGL(k_pack_wh, dim3(16, (unsigned)t->pack_tab_h.size()), dim3(256), 0, t->eng->stream, t->w, t->wpack, t->wsc,
                       t->pack_dev_h);
}
// Y[rows x NO] = X[rows x NR] * Wop (+ epilogue) on k_lin32 / k_lin32f; a.Wp is filled in here
// b / Wb: a second product of the same shape (rows, NR, NO) and epilogue class to run in the same launch (k_lin32h only:
// the other kernels get two launches); an accumulating job may pair with a storing one (both run the accumulate variant)
static int lin32(lb_gns_train* t, lb_lin_args a, const float* W, int ldw, int trans, const lb_lin_args* b = nullptr,
                 const float* Wb = nullptr) {
  if (a.rows == 0) return LB_OK;
  if (b) {
    const bool same = b->rows == a.rows && b->NR == a.NR && b->NO == a.NO && b->ldx == a.ldx && b->ldy == a.ldy &&
                      !a.mask && !b->mask && !a.ln_scale && !b->ln_scale && !a.gat1 && !b->gat1 && !a.tmax && !b->tmax &&
                      !a.bias == !b->bias && a.relu == b->relu;
    const bool fast_pair = a.NO == 128 && (a.NR & 127) == 0 && (a.ldx & 3) == 0 && (a.ldy & 3) == 0 &&
                           (((uintptr_t)a.bias | (uintptr_t)b->bias) & 15) == 0;
#ifdef LB_NO_PAIR   // (tools/build_variant.sh A/B: every product a launch of its own)
    const bool pair_ok = false;
#else
    const bool pair_ok = true;
#endif
    if (!(pair_ok && same && fast_pair && t->f16x2)) {
      LB_TRY(lin32(t, a, W, ldw, trans));
      return lin32(t, *b, Wb, ldw, trans);
    }
  }
  if (a.NR > 256 || a.NO > 128) return lb_fail(LB_ERR_UNSUPPORTED, "k_lin32: %d x %d operand", a.NR, a.NO);
  const int nob = a.NO <= 16 ? 1 : (a.NO <= 64 ? 4 : 8);  // 16-column output blocks per wave (generic kernel)
  a.NJ = (a.NR + 15) / 16;
  size_t lds = (size_t)a.NJ * nob * 64 * sizeof(f32x4);
  const int64_t tiles = (a.rows + 15) / 16;
  const int grid = (int)std::min<int64_t>(tiles, 256);  // one workgroup per CU; tile t -> workgroup t % grid first
  hipStream_t s = t->eng->stream;
  const bool fast = a.NO == 128 && (a.NR & 127) == 0 && (a.ldx & 3) == 0 && (a.ldy & 3) == 0 && (!a.mask || (a.ldm & 3) == 0) &&
                    !(a.mask && a.accum) && (((uintptr_t)a.bias | (uintptr_t)a.ln_scale | (uintptr_t)a.ln_offset) & 15) == 0;
  const bool half = fast && t->f16x2;
  if ((trans == 2 || a.xcs) && !half) return lb_fail(LB_ERR_UNSUPPORTED, "k_lin32: stacked operands run on k_lin32h only");
  t->tmax_ok = half && a.tmax;
  if (!half) a.tmax = nullptr;
  if (half) LB_TRY(pack_lookup_h(t, W, a.NR, a.NO, ldw, trans, &a.Wp, &a.wsc));
  else LB_TRY(pack_lookup(t, W, a.NR, a.NO, ldw, trans, nob, &a.Wp));
  lb_lin_args b2 = a;   // (single launches pass their own job twice: blockIdx.y is 0)
  if (b) {
    b2 = *b;
    b2.NJ = a.NJ;
    LB_TRY(pack_lookup_h(t, Wb, b2.NR, b2.NO, ldw, trans, &b2.Wp, &b2.wsc));
  }
  const unsigned gy = b ? 2u : 1u;
  const bool pair_accum = b && (a.accum || b->accum);
  if (a.gat1 && (!fast || a.mask || a.accum || a.ln_scale || !a.bias))
    return lb_fail(LB_ERR_UNSUPPORTED, "k_lin32: gather epilogue on a %d x %d operand", a.NR, a.NO);
  if (a.ln_scale && (!fast || a.mask || a.accum || a.relu))
    return lb_fail(LB_ERR_UNSUPPORTED, "k_lin32: LayerNorm epilogue on a %d x %d operand", a.NR, a.NO);
  if (fast) lds += 96 * sizeof(f32x4);  // bias | LayerNorm scale | offset
#define LB_LIN_GO(KERNEL)                                                                                                    \
  do {                                                                                                                       \
    static bool raised = false;                                                                                              \
    if (!raised) {                                                                                                           \
      (void)hipFuncSetAttribute((const void*)KERNEL, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);                \
      raised = true;                                                                                                         \
    }                                                                                                                        \
    hipLaunchKernelGGL(KERNEL, dim3(grid), dim3(512), lds, s, a);                                                            \
  } while (0)
#define LB_LIN_GO2(KERNEL)                                                                                                   \
  do {                                                                                                                       \
    static bool raised = false;                                                                                              \
    if (!raised) {                                                                                                           \
      (void)hipFuncSetAttribute((const void*)KERNEL, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);                \
      raised = true;                                                                                                         \
    }                                                                                                                        \
    hipLaunchKernelGGL(KERNEL, dim3(grid, gy), dim3(512), lds, s, a, b2);                                                    \
  } while (0)
  if (half) {
    if (a.gat1) LB_LIN_GO2(k_lin32h<4>);
    else if (a.ln_scale) LB_LIN_GO2(k_lin32h<3>);
    else if (a.mask) LB_LIN_GO2(k_lin32h<1>);
    else if (a.accum || pair_accum) LB_LIN_GO2(k_lin32h<2>);
    else LB_LIN_GO2(k_lin32h<0>);
  } else if (fast) {
    if (a.gat1) LB_LIN_GO(k_lin32f<4>);
    else if (a.ln_scale) LB_LIN_GO(k_lin32f<3>);
    else if (a.mask) LB_LIN_GO(k_lin32f<1>);
    else if (a.accum) LB_LIN_GO(k_lin32f<2>);
    else LB_LIN_GO(k_lin32f<0>);
  } else if (nob == 1) LB_LIN_GO(k_lin32<1>);
  else if (nob == 4) LB_LIN_GO(k_lin32<4>);
  else LB_LIN_GO(k_lin32<8>);
#undef LB_LIN_GO
#undef LB_LIN_GO2
  LB_HIP(hipGetLastError());
  return LB_OK;
}
// row-major Y[rows x M] (ldy) = X[rows x K] (ldx) * W[K x M] (+ beta * Y) [+ bias, ReLU]
struct lb_ln_epi {  // LayerNorm epilogue of gemm_nn (k_lin32f<3>): see lb_lin_args
  const float *scale, *offset, *resid;
  float *yln, *y2;
  int d;
};
static int gemm_nn(lb_gns_train* t, int64_t rows, int M, int K, const float* X, int ldx, const float* W, float* Y,
                   int ldy, float beta = 0.f, const float* bias = nullptr, int relu = 0, const lb_ln_epi* ln = nullptr) {
  lb_lin_args a{};
  a.X = X; a.ldx = ldx; a.NR = K; a.Y = Y; a.ldy = ldy; a.NO = M; a.rows = rows;
  a.bias = bias; a.relu = relu; a.accum = beta != 0.f;
  if (ln) {
    a.ln_scale = ln->scale; a.ln_offset = ln->offset; a.resid = ln->resid; a.Yln = ln->yln; a.Y2 = ln->y2; a.ln_d = ln->d;
  }
  return lin32(t, a, W, M, 0);
}
// dX[rows x K] (ldx) = dY[rows x M] (ldy) * W^T (+ beta * dX) [* (mask > 0)]; K > 128: 128 output columns per launch
static int gemm_nt(lb_gns_train* t, int64_t rows, int M, int K, const float* dY, const float* W, float* dX, int ldx,
                   float beta = 0.f, int ldy = 0, const float* mask = nullptr, int ldm = 0, float* tmax = nullptr) {
  for (int c0 = 0; c0 < K; c0 += 128) {
    lb_lin_args a{};
    a.tmax = tmax;
    a.X = dY; a.ldx = ldy ? ldy : M; a.NR = M; a.Y = dX + c0; a.ldy = ldx; a.NO = std::min(128, K - c0); a.rows = rows;
    a.mask = mask ? mask + c0 : nullptr; a.ldm = ldm; a.accum = beta != 0.f;
    LB_TRY(lin32(t, a, W + (size_t)c0 * M, M, 1));
  }
  return LB_OK;
}
// two products of one shape in ONE launch (k_lin32h's paired form; lin32 falls back to two launches where it does not apply):
// Ya = X Wa, Yb = X Wb (K x 128 operands, no epilogue) ...
static int gemm_nn_pair(lb_gns_train* t, int64_t rows, int K, const float* X, int ldx, const float* Wa, float* Ya, const float* Wb,
                        float* Yb, int ldy) {
  lb_lin_args a{};
  a.X = X; a.ldx = ldx; a.NR = K; a.Y = Ya; a.ldy = ldy; a.NO = TD; a.rows = rows;
  lb_lin_args b = a;
  b.Y = Yb;
  return lin32(t, a, Wa, TD, 0, &b, Wb);
}
// ... and dXa = dY Wa^T (+ beta_a dXa), dXb = dY Wb^T (+ beta_b dXb) for 128 x 128 operands
static int gemm_nt_pair(lb_gns_train* t, int64_t rows, const float* dY, const float* Wa, float* dXa, float beta_a, const float* Wb,
                        float* dXb, float beta_b) {
  lb_lin_args a{};
  a.X = dY; a.ldx = TD; a.NR = TD; a.Y = dXa; a.ldy = TD; a.NO = TD; a.rows = rows; a.accum = beta_a != 0.f;
  lb_lin_args b = a;
  b.Y = dXb; b.accum = beta_b != 0.f;
  return lin32(t, a, Wa, TD, 1, &b, Wb);
}
// dX += dYa Wa^T + dYb Wb^T for two 128 x 128 operands that are consecutive row blocks of one matrix (Wb = Wa + 128 * 128)
// and two dY arrays `gap` floats apart: ONE product with a 256-deep reduction (k_lin32h reads its two chunks from the two
// arrays, k_pack_wh stacks the two transposes: trans = 2) - the two accumulating launches it replaces were serialised by dX
static int gemm_nt_stack2(lb_gns_train* t, int64_t rows, const float* dYa, int64_t gap, const float* Wa, float* dX) {
#ifndef LB_NO_PAIR
  if (t->f16x2) {
    lb_lin_args a{};
    a.X = dYa; a.ldx = TD; a.NR = 2 * TD; a.xcs = gap; a.Y = dX; a.ldy = TD; a.NO = TD; a.rows = rows; a.accum = 1;
    return lin32(t, a, Wa, TD, 2);
  }
#endif
  LB_TRY(gemm_nt(t, rows, TD, TD, dYa, Wa, dX, TD, 1.f));
  return gemm_nt(t, rows, TD, TD, dYa + gap, Wa + (size_t)TD * TD, dX, TD, 1.f);
}
// ---- the step's gradient reductions: producers take a slot for their partials and leave a descriptor, red_flush sums all
// of them in one launch (lb_red_ent / k_part_reduce)
#define DW_MAX_G 256
#define LB_RED_MAX 512
static int dw_groups(int64_t rows, int64_t* chunk_out) {
  int64_t chunk = (rows + DW_MAX_G - 1) / DW_MAX_G;
  if (chunk < 64) chunk = 64;  // (128 halves the partials of node-sized products but k_dw_part is a latency chain per
                               //  workgroup: 10 -> 16 us per launch on TGV2D, measured)
  chunk = (chunk + 3) / 4 * 4;
  if (chunk_out) *chunk_out = chunk;
  return (int)((rows + chunk - 1) / chunk);
}
// upper bound of dw_groups over every row count <= rows_cap (dw_groups is not monotonic: the chunk is rounded up to 4 rows)
static int64_t dw_groups_max(int64_t rows_cap) { return std::min<int64_t>(DW_MAX_G, (rows_cap + 63) / 64 + 1); }
static float* red_slot(lb_gns_train* t, int64_t floats, int64_t* off) {
  floats = (floats + 63) / 64 * 64;
  if (t->red_off + floats > t->red_cap || t->red_tab.size() >= LB_RED_MAX) {
    lb_fail(LB_ERR_STATE, "training: the partial-sum buffer is full (%lld + %lld of %lld floats, %zu reductions)",
            (long long)t->red_off, (long long)floats, (long long)t->red_cap, t->red_tab.size());
    return nullptr;
  }
  *off = t->red_off;
  t->red_off += floats;
  return t->dwpart + *off;
}
static void red_push(lb_gns_train* t, int64_t part, int G, int64_t stride, int n0, int n1, int off1, const float* dst0,
                     const float* dst1) {
  lb_red_ent d{};
  d.part = part; d.stride = stride; d.G = G; d.n0 = n0; d.n1 = n1; d.off1 = off1;
  d.dst0 = dst0 - t->g;
  d.dst1 = dst1 ? dst1 - t->g : 0;
  d.blk0 = t->red_blocks;
  t->red_tab.push_back(d);
  t->red_blocks += (n0 + n1 + RED_OUT - 1) / RED_OUT;
}
static int red_flush(lb_gns_train* t) {
  const size_t n = t->red_tab.size();
  if (n) {
    hipStream_t s = t->eng->stream;
    memcpy(t->red_host, t->red_tab.data(), n * sizeof(lb_red_ent));  // (pinned; the previous step's copy was synchronised)
    LB_HIP(hipMemcpyAsync(t->red_dev, t->red_host, n * sizeof(lb_red_ent), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_part_reduce, dim3((unsigned)t->red_blocks), dim3(1024), 0, s, t->dwpart, t->red_dev, (int)n, t->g,
                       t->dw_flag);
  }
  t->red_tab.clear();
  t->red_off = 0;
  t->red_blocks = 0;
  return LB_OK;
}
// floats of partial-sum slots a step needs for BN <= cn nodes and E <= ce edges (mirrors the backward pass below)
static int64_t red_capacity(const lb_gns_train* t, int64_t cn, int64_t ce) {
  auto slot = [](int64_t rows, int K) { return (dw_groups_max(std::max<int64_t>(rows, 1)) * (K + 1) * 128 + 63) / 64 * 64; };
  auto lnp = [](int64_t rows) { return ((rows + LNB_ROWS - 1) / LNB_ROWS * 256 + 63) / 64 * 64 + 64; };
  const int L = t->desc.num_mp_steps;
  int64_t tot = 0;
  tot += (int64_t)L * (2 * slot(ce, 128) + 3 * slot(cn, 128) + slot(cn, 256) + lnp(ce) + lnp(cn));  // processor blocks
  tot += slot(cn, 128) + slot(cn, std::max(128, t->kpad)) + lnp(cn);                             // node encoder
  tot += 2 * slot(ce, 128) + lnp(ce);                                                            // edge encoder
  tot += 2 * slot(cn, 128) + (cn + 127) / 128 * 128 + 64;                                        // decoder (narrow dW, column sums)
  const int64_t mid = t->nlin - 2;                                                               // middle Linears of every block
  tot += mid * ((int64_t)L * (slot(ce, 128) + slot(cn, 128)) + 2 * slot(cn, 128) + slot(ce, 128));
  return tot + 4096;
}
// dW[K x 128] += X^T dY, db[128] += column sums of dY (k_dw_part + a descriptor for k_part_reduce); false = refused
// (nb: how many of dY's 128 column sums are added to db - the SEGNN blocks keep only their Ms scalar-output columns)
// dY_b / dW_b (dw_acc_pair): a second product X^T dY_b -> dW_b (no bias) in the same launch - f16x2 kernel only
static bool dw_acc(lb_gns_train* t, int64_t rows, int K, const float* X, int ldx, const float* dY, float* dW, float* db,
                   int nb, const float* tmax, const float* dY_b, float* dW_b) {
  // (every refusal leaves its reason in lb_last_error - ADVICE r05: callers return a bare LB_ERR_STATE)
  if (K > 384 || rows <= 0) {
    (void)lb_fail(LB_ERR_STATE, "dw_acc: K = %d (<= 384) / rows = %lld (> 0) out of range", K, (long long)rows);
    return false;
  }
  int64_t chunk = 0, off = 0;
  const int G = dw_groups(rows, &chunk);
  hipStream_t s = t->eng->stream;
  if ((K & 1) && ldx <= K) {  // (the pair load of an odd K reads the row's padding column: never stored)
    (void)lb_fail(LB_ERR_STATE, "dw_acc: odd K = %d needs a padded row (ldx = %d)", K, ldx);
    return false;
  }
  float* part = red_slot(t, (int64_t)G * (K + 1) * 128, &off);
  if (!part) return false;  // (red_slot said why)
  int64_t off_b = 0;
  float* part_b = nullptr;
  if (dY_b) {
    if (!(t->f16x2 && K >= 32)) {
      (void)lb_fail(LB_ERR_STATE, "dw_acc: paired products run on the f16x2 kernel only");
      return false;
    }
    part_b = red_slot(t, (int64_t)G * (K + 1) * 128, &off_b);
    if (!part_b) return false;
  }
#define DW_GO(NA, DEPTH) hipLaunchKernelGGL((k_dw_part<NA, DEPTH>), dim3(G), dim3(512), 0, s, X, ldx, K, dY, rows, chunk, part)
  const bool deep = chunk >= 192;
  if (t->f16x2 && K >= 32) {   // (narrower operands - the encoders' raw features - stay on the fp32 kernel)
    const int c = std::min(t->dw_call++, LB_DW_CALLS - 1);
    if ((int)t->dw_xexp.size() <= c) { t->dw_xexp.resize(c + 1, 0); t->dw_xdyn.resize(c + 1, 0); }
    hipLaunchKernelGGL(k_dw_part_h, dim3((unsigned)G, (unsigned)((K + 127) / 128), part_b ? 2u : 1u), dim3(512), 0, s, X, ldx, K, dY,
                       rows, chunk, part, tmax, t->dw_flag, t->dw_flag + 1 + c, (int)t->dw_xexp[c], (int)t->dw_xdyn[c], dY_b, part_b);
  }
  else if (K <= 128) { if (deep) DW_GO(1, 12); else DW_GO(1, 6); }
  else if (K <= 256) { if (deep) DW_GO(2, 8); else DW_GO(2, 6); }
  else DW_GO(3, 4);
#undef DW_GO
  red_push(t, off, G, (int64_t)(K + 1) * 128, K * 128, db ? nb : 0, K * 128, dW, db);
  if (part_b) red_push(t, off_b, G, (int64_t)(K + 1) * 128, K * 128, 0, K * 128, dW_b, nullptr);
  return true;
}
// dW_a += X^T dY_a and dW_b += X^T dY_b: one launch on the f16x2 kernel, two otherwise
static bool dw_acc_pair(lb_gns_train* t, int64_t rows, int K, const float* X, int ldx, const float* dY_a, float* dW_a,
                        const float* dY_b, float* dW_b) {
#ifndef LB_NO_PAIR
  if (t->f16x2 && K >= 32) return dw_acc(t, rows, K, X, ldx, dY_a, dW_a, nullptr, 128, nullptr, dY_b, dW_b);
#endif
  return dw_acc(t, rows, K, X, ldx, dY_a, dW_a, nullptr, 128, nullptr, nullptr, nullptr) &&
         dw_acc(t, rows, K, X, ldx, dY_b, dW_b, nullptr, 128, nullptr, nullptr, nullptr);
}
// dW[K x M] += X^T dY for M <= 4 (k_dw_narrow + the ordered reduce)
static int dw_narrow(lb_gns_train* t, int64_t rows, int M, int K, const float* X, int ldx, const float* dY, int ldy, float* dW) {
  if (rows == 0) return LB_OK;
  if (M > 4 || K > 128) return lb_fail(LB_ERR_UNSUPPORTED, "dw_narrow: %d x %d", K, M);
  int64_t chunk = (rows + DW_MAX_G - 1) / DW_MAX_G, off = 0;
  if (chunk < 64) chunk = 64;
  const int G = (int)((rows + chunk - 1) / chunk);
  float* part = red_slot(t, (int64_t)G * K * M, &off);
  if (!part) return LB_ERR_STATE;
  hipLaunchKernelGGL(k_dw_narrow, dim3(G), dim3(128), 0, t->eng->stream, X, ldx, K, dY, ldy, M, rows, chunk, part);
  red_push(t, off, G, (int64_t)K * M, K * M, 0, 0, dW, nullptr);
  return LB_OK;
}
// out[cols] += column sums of x (k_colsum_part: fixed blocks of 128 rows, summed in block order by k_part_reduce)
static int colsum_add(lb_gns_train* t, const float* x, int64_t rows, int cols, int ld, float* out) {
  if (rows == 0) return LB_OK;
  const int nb = (int)((rows + 127) / 128);
  int64_t off = 0;
  float* part = red_slot(t, (int64_t)nb * 128, &off);
  if (!part) return LB_ERR_STATE;
  hipLaunchKernelGGL(k_colsum_part, dim3(nb), dim3(128), 0, t->eng->stream, x, rows, cols, ld, part);
  red_push(t, off, nb, 128, cols, 0, 0, out, nullptr);
  return LB_OK;
}

static int mlp_fwd_tail(lb_gns_train* t, const lb_train_mlp& p, int64_t rows, float* a, float* z, const float* resid, float* y,
                        float* y2);
// forward of one MLP block: X (rows x in, ldx) -> a = relu(X W0 + b0) -> z = a W1 + b1 -> [LayerNorm (+ resid)] -> y
// the middle Linears of a block: a_{m+1} = relu(a_m Wm + bm), hidden m at a + m * stride
static int mlp_mid_fwd(lb_gns_train* t, const lb_train_mlp& p, int64_t rows, float* a, int64_t stride) {
  for (size_t m = 0; m < p.wm.size(); ++m)
    LB_TRY(gemm_nn(t, rows, TD, TD, a + m * stride, TD, t->w + p.wm[m], a + (m + 1) * stride, TD, 0.f, t->w + p.bm[m], 1));
  return LB_OK;
}
// ... and their backward.  In: t->da = d(last hidden), ReLU-masked; out: t->da = d(first hidden), masked (the two buffers swap)
static bool dw_acc(lb_gns_train* t, int64_t rows, int K, const float* X, int ldx, const float* dY, float* dW, float* db,
                   int nb = 128, const float* tmax = nullptr, const float* dY_b = nullptr, float* dW_b = nullptr);
static bool dw_acc_pair(lb_gns_train* t, int64_t rows, int K, const float* X, int ldx, const float* dY_a, float* dW_a,
                        const float* dY_b, float* dW_b);
static int mlp_mid_bwd(lb_gns_train* t, const lb_train_mlp& p, int64_t rows, const float* a, int64_t stride) {
  for (size_t m = p.wm.size(); m-- > 0;) {
    const float* ain = a + m * stride;  // input of middle Linear m
    LB_TRY(gemm_nt(t, rows, TD, TD, t->da, t->w + p.wm[m], t->da2, TD, 0.f, 0, ain, TD, t->tmax));  // ReLU mask of ain
    if (!dw_acc(t, rows, TD, ain, TD, t->da, t->g + p.wm[m], t->g + p.bm[m], 128, t->tmax_ok ? t->tmax : nullptr))
      return lb_fail(LB_ERR_STATE, "dw_acc refused a middle Linear");
    std::swap(t->da, t->da2);
  }
  t->tmax_ok = false;
  return LB_OK;
}
static int mlp_fwd(lb_gns_train* t, const lb_train_mlp& p, int64_t rows, const float* X, int ldx, float* a, float* z,
                   const float* resid, float* y, int64_t stride) {
  hipStream_t s = t->eng->stream;
  (void)s;
  LB_TRY(gemm_nn(t, rows, TD, p.in, X, ldx, t->w + p.w0, a, TD, 0.f, t->w + p.b0, 1));  // bias + ReLU in the epilogue
  LB_TRY(mlp_mid_fwd(t, p, rows, a, stride));
  float* al = a + p.wm.size() * stride;
  return resid ? mlp_fwd_tail(t, p, rows, al, z, resid, nullptr, y) : mlp_fwd_tail(t, p, rows, al, z, nullptr, y, nullptr);
}
// forward of the edge block (gns.py:86-101) without the concatenated input (see k_edge_dP's comment)
// y = e' = LN(MLP([n_s | n_r | e])) (the message), el_next = e + e' (the next edge latent)
static int edge_fwd(lb_gns_train* t, const lb_train_mlp& p, int64_t E, int64_t BN, const float* n, const float* el, float* a,
                    float* z, float* y, float* el_next) {
  hipStream_t s = t->eng->stream;
  lb_engine* e = t->eng;
  float *Ps = t->proj, *Pr = t->proj + (size_t)BN * TD;
  LB_TRY(gemm_nn_pair(t, BN, TD, n, TD, t->w + p.w0, Ps, t->w + p.w0 + (size_t)TD * TD, Pr, TD));
  {  // a = relu(e W_e + Ps[snd] + Pr[rcv] + b0): the gathers and the bias ride in the product's epilogue
    lb_lin_args g{};
    g.X = el; g.ldx = TD; g.NR = TD; g.Y = a; g.ldy = TD; g.NO = TD; g.rows = E;
    g.bias = t->w + p.b0; g.gat1 = Ps; g.gidx1 = e->senders; g.gat2 = Pr; g.gidx2 = e->receivers;
    LB_TRY(lin32(t, g, t->w + p.w0 + (size_t)2 * TD * TD, TD, 0));
  }
  LB_TRY(mlp_mid_fwd(t, p, E, a, t->hs_e));
  return mlp_fwd_tail(t, p, E, a + p.wm.size() * t->hs_e, z, el, y, el_next);
}
// y = LN(...) (may be null), y2 = resid + y (may be null); without LayerNorm (decoder): y = a W1 + b1
static int mlp_fwd_tail(lb_gns_train* t, const lb_train_mlp& p, int64_t rows, float* a, float* z, const float* resid,
                        float* y, float* y2) {
  hipStream_t s = t->eng->stream;
  (void)s;
  if (!p.ln) return gemm_nn(t, rows, p.out, TD, a, TD, t->w + p.w1, y, p.out, 0.f, t->w + p.b1, 0);
  // second Linear with the LayerNorm in its epilogue: z = a W1 (kept without the bias for the backward), y = LN(z + b1),
  // y2 = resid + y (round 4: k_ln_fwd and, in the edge block, k_add2 were launches of their own over the same arrays)
  lb_ln_epi ln{t->w + p.lns, t->w + p.lno, resid, y, y2, t->lat};
  return gemm_nn(t, rows, TD, TD, a, TD, t->w + p.w1, z, TD, 0.f, t->w + p.b1, 0, &ln);
}
// backward of one MLP block.  dy: gradient w.r.t. the block's output BEFORE the residual add (rows x out);
// produces parameter gradients (accumulated) and, if dX != null, dX (rows x in, ld = ldx).  Scratch: t->dz, t->da.
// first part (shared with the edge block): LayerNorm and second Linear backward, ReLU mask -> t->da = d loss / d (X W0 + b0)
static int mlp_bwd_head(lb_gns_train* t, const lb_train_mlp& p, int64_t rows, const float* a, const float* z, const float* dy,
                        const float* gth = nullptr, const int32_t* gidx = nullptr) {
  hipStream_t s = t->eng->stream;
  if (rows == 0) return LB_OK;
  const float* dzz = dy;
  if (p.ln) {
    const int nb = (int)((rows + LNB_ROWS - 1) / LNB_ROWS);
    int64_t off = 0;
    float* part = red_slot(t, (int64_t)nb * 256, &off);
    if (!part) return LB_ERR_STATE;
    if (gth)
      hipLaunchKernelGGL(k_ln_bwd2<true>, dim3(nb), dim3(256), 0, s, z, t->w + p.b1, t->w + p.lns, dy, t->dz, rows, part, t->lat, gth, gidx);
    else
      hipLaunchKernelGGL(k_ln_bwd2<false>, dim3(nb), dim3(256), 0, s, z, t->w + p.b1, t->w + p.lns, dy, t->dz, rows, part, t->lat, gth, gidx);
    red_push(t, off, nb, 256, 128, 128, 128, t->g + p.lns, t->g + p.lno);
    dzz = t->dz;
  }
  // dX first: k_lin32h leaves the row-tile maxima of dzz behind, which the weight-gradient kernel scales by
  LB_TRY(gemm_nt(t, rows, p.out, TD, dzz, t->w + p.w1, t->da, TD, 0.f, 0, a, TD, t->tmax));  // ReLU mask in the epilogue
  if (p.out != TD || !dw_acc(t, rows, TD, a, TD, dzz, t->g + p.w1, t->g + p.b1, 128, t->tmax_ok ? t->tmax : nullptr)) {
    LB_TRY(dw_narrow(t, rows, p.out, TD, a, TD, dzz, p.out, t->g + p.w1));
    LB_TRY(colsum_add(t, dzz, rows, p.out, p.out, t->g + p.b1));
  }
  return LB_OK;
}
static int mlp_bwd(lb_gns_train* t, const lb_train_mlp& p, int64_t rows, const float* X, int ldx, const float* a,
                   const float* z, const float* dy, float* dX, int64_t stride) {
  if (rows == 0) return LB_OK;
  LB_TRY(mlp_bwd_head(t, p, rows, a + p.wm.size() * stride, z, dy));
  LB_TRY(mlp_mid_bwd(t, p, rows, a, stride));
  t->tmax_ok = false;
  if (dX) LB_TRY(gemm_nt(t, rows, TD, p.in, t->da, t->w + p.w0, dX, ldx, 0.f, 0, nullptr, 0, t->tmax));
  if (!dw_acc(t, rows, p.in, X, ldx, t->da, t->g + p.w0, t->g + p.b0, 128, t->tmax_ok ? t->tmax : nullptr))
    return lb_fail(LB_ERR_UNSUPPORTED, "training: first Linear with %d inputs (row stride %d) is not covered", p.in, ldx);
  return LB_OK;
}

// backward of the edge block: parameter gradients, de += d loss / d e_k, dn += the node-side terms (see k_edge_dP)
static int edge_bwd(lb_gns_train* t, const lb_train_mlp& p, int64_t E, int64_t BN, const float* n, const float* el,
                    const float* a, const float* z, const float* dagg, float* de, float* dn) {
  if (E == 0) return LB_OK;
  hipStream_t s = t->eng->stream;
  lb_engine* e = t->eng;
  // e' feeds agg (gather of dagg over the receivers) and e_{k+1} = e_k + e' (de): d e' = de + dagg[rcv], formed in the load
  LB_TRY(mlp_bwd_head(t, p, E, a + p.wm.size() * t->hs_e, z, de, dagg, e->receivers));  // -> t->da (E x 128)
  LB_TRY(mlp_mid_bwd(t, p, E, a, t->hs_e));
  const float *Ws = t->w + p.w0, *Wr = Ws + (size_t)TD * TD, *We = Wr + (size_t)TD * TD;
  float *gWs = t->g + p.w0, *gWr = gWs + (size_t)TD * TD, *gWe = gWr + (size_t)TD * TD;
  // edge rows: dW_e += e^T da, db0 += column sums of da, de += da W_e^T
  LB_TRY(gemm_nt(t, E, TD, TD, t->da, We, de, TD, 1.f, 0, nullptr, 0, t->tmax));
  if (!dw_acc(t, E, TD, el, TD, t->da, gWe, t->g + p.b0, 128, t->tmax_ok ? t->tmax : nullptr))
    return lb_fail(LB_ERR_STATE, "dw_acc refused the edge block");
  // node rows: dP = transpose of the two gathers, then dW_s += n^T dPs, dW_r += n^T dPr, dn += dPs W_s^T + dPr W_r^T
  float *dPs = t->proj, *dPr = t->proj + (size_t)BN * TD;
  hipLaunchKernelGGL(k_edge_dP, GRID1(BN * 32), 0, s, t->da, t->snd_ptr, t->snd_perm, e->row_ptr, dPs, dPr, BN, E);
  if (!dw_acc_pair(t, BN, TD, n, TD, dPs, gWs, dPr, gWr)) return lb_fail(LB_ERR_STATE, "dw_acc refused the edge block");
  (void)Wr;
  return gemm_nt_stack2(t, BN, dPs, (int64_t)BN * TD, Ws, dn);
}

template <typename T>
static int tr_alloc(T** p, size_t n) {
  if (*p) (void)hipFree(*p);
  return lb_alloc(p, n);
}

static int train_ensure(lb_gns_train* t, int64_t BN, int64_t E) {
  const int L = t->desc.num_mp_steps;
  if (BN <= t->cap_n && E <= t->cap_e && t->xnode) return LB_OK;
  LB_HIP(hipStreamSynchronize(t->eng->stream));
  const int64_t cn = std::max(BN, t->cap_n), ce = std::max(E + E / 8 + 1024, t->cap_e);
  const size_t nh = (size_t)(t->nlin - 1);   // hidden activations kept per block
  t->hs_n = cn * TD;
  t->hs_e = ce * TD;
  LB_TRY(tr_alloc(&t->xnode, (size_t)cn * t->kpad));
  LB_TRY(tr_alloc(&t->a_en, nh * cn * TD));
  LB_TRY(tr_alloc(&t->z_en, (size_t)cn * TD));
  LB_TRY(tr_alloc(&t->a_ee, nh * ce * TD));
  LB_TRY(tr_alloc(&t->z_ee, (size_t)ce * TD));
  for (int k = 0; k <= L; ++k) {
    LB_TRY(tr_alloc(&t->nlat[k], (size_t)cn * TD));
    LB_TRY(tr_alloc(&t->elat[k], (size_t)ce * TD));
  }
  for (int k = 0; k < L; ++k) {
    LB_TRY(tr_alloc(&t->ae[k], nh * ce * TD));
    LB_TRY(tr_alloc(&t->ze[k], (size_t)ce * TD));
    LB_TRY(tr_alloc(&t->xn[k], (size_t)cn * 2 * TD));
    LB_TRY(tr_alloc(&t->an[k], nh * cn * TD));
    LB_TRY(tr_alloc(&t->zn[k], (size_t)cn * TD));
  }
  const int64_t cm = std::max(cn, ce);
  LB_TRY(tr_alloc(&t->a_d, nh * cn * TD));
  LB_TRY(tr_alloc(&t->pred, (size_t)cn * 4));
  LB_TRY(tr_alloc(&t->dn, (size_t)cn * TD));
  LB_TRY(tr_alloc(&t->de, (size_t)ce * TD));
  LB_TRY(tr_alloc(&t->dy, (size_t)cm * TD));
  LB_TRY(tr_alloc(&t->dz, (size_t)cm * TD));
  LB_TRY(tr_alloc(&t->da, (size_t)cm * TD));
  if (t->nlin > 2) LB_TRY(tr_alloc(&t->da2, (size_t)cm * TD));
  LB_TRY(tr_alloc(&t->dx, (size_t)cn * std::max(3 * TD, t->kpad)));
  LB_TRY(tr_alloc(&t->dagg, (size_t)cn * TD));
  LB_TRY(tr_alloc(&t->agg, (size_t)cn * TD));
  LB_TRY(tr_alloc(&t->tmax, (size_t)(cm / 16 + 64)));
  t->red_cap = red_capacity(t, cn, ce);
  LB_TRY(tr_alloc(&t->dwpart, (size_t)t->red_cap));
  if (!t->red_dev) {
    LB_TRY(lb_alloc(&t->red_dev, (size_t)LB_RED_MAX));
    LB_HIP(hipHostMalloc((void**)&t->red_host, sizeof(lb_red_ent) * (LB_RED_MAX + 1)));   // (+ the step's status words)
  }
  LB_TRY(tr_alloc(&t->proj, (size_t)cn * 2 * TD));
  LB_TRY(tr_alloc(&t->node_w, (size_t)cn));
  LB_TRY(tr_alloc(&t->loss_part, (size_t)(cn / 64 + 8)));
  t->cap_n = cn;
  t->cap_e = ce;
  return LB_OK;
}

// _mse of trainer.py:35-60 on the device: loss (into t->loss_dev) and d loss / d pred (BN x dim) for the whole batch
static int train_loss(lb_gns_train* t, const float* pred, const float* target_dev, float loss_weight, float* dpred) {
  lb_engine* e = t->eng;
  hipStream_t s = e->stream;
  const int64_t BN = e->BN;
  LB_HIP(hipMemsetAsync(t->cnt_dev, 0, sizeof(int32_t) * e->g.B, s));
  LB_HIP(hipMemsetAsync(t->loss_dev, 0, sizeof(double), s));
  hipLaunchKernelGGL(k_count_nonkin, GRID1(BN), 0, s, e->ptype, BN, e->g.N, t->cnt_dev);
  hipLaunchKernelGGL(k_node_weight, GRID1(BN), 0, s, e->ptype, t->cnt_dev, BN, e->g.N, t->node_w);
  hipLaunchKernelGGL(k_mse_grad, GRID1(BN), 0, s, pred, target_dev, t->node_w, BN, e->g.dim, loss_weight, 1.0f / (float)e->g.B,
                     dpred, t->loss_part);
  hipLaunchKernelGGL(k_loss_finish, dim3(1), dim3(64), 0, s, t->loss_part, (int64_t)((BN + 255) / 256) * 4, t->loss_dev);
  return LB_OK;
}
// Sender-sorted view of the step's edge list WITHOUT a sort (round 6; hipcub::DeviceRadixSort until then: three rocprim
// kernels + a lower-bound pass per step).  The neighbor relation is symmetric - (r, s) is an edge iff (s, r) is - and the list
// is sorted by (receiver, sender): the edges SENT by node s are the transposes of row s, so s sends as many edges as it
// receives (snd_ptr == row_ptr) and, among the edges sent by s in ascending edge order (= ascending receiver), edge (r, s)
// has the rank that r has among the senders of row s.  One thread per edge: a bisection of row s (a dozen L2-resident
// integers) for r, then snd_perm[row_ptr[s] + rank] = edge.  The periodic displacement is antisymmetric only up to one
// rounding of (x + L/2), so a pair within ~1e-16 of the cutoff CAN be an edge in one direction only: a thread that does
// not find its transpose raises bit 4 of the step's guard flag - the reductions then add nothing and the step is repeated
// with the radix sort below (train_loss_grad_guarded).
__global__ void k_sender_transpose(const int32_t* __restrict__ row_ptr, const int32_t* __restrict__ senders,
                                   const int32_t* __restrict__ receivers, int64_t E, int64_t BN, int32_t* __restrict__ snd_perm,
                                   int32_t* __restrict__ snd_ptr, int32_t* __restrict__ flag) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i <= BN) snd_ptr[i] = row_ptr[i];
  if (i >= E) return;
  const int r = receivers[i], s = senders[i];
  int lo = row_ptr[s], hi = row_ptr[s + 1];   // first position of row s whose sender is >= r
  const int end = hi;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (senders[mid] < r) lo = mid + 1; else hi = mid;
  }
  if (lo < end && senders[lo] == r)
    snd_perm[lo] = (int32_t)i;
  else
    atomicOr(flag, 4);
}
// the same view by a stable radix sort of (sender, edge index) - the fall-back for a step whose list is not symmetric
// -> t->snd_perm (E), t->snd_ptr (BN + 1)
static int train_sender_sort(lb_gns_train* t, int64_t E, int64_t BN) {
  lb_engine* e = t->eng;
  hipStream_t s = e->stream;
  if (!E) return LB_OK;
  static const bool force_cub = getenv("LB_TRAIN_SORT") && getenv("LB_TRAIN_SORT")[0] == 'c';  // LB_TRAIN_SORT=cub: always the fall-back (test)
  if (!t->cub_sort && !force_cub) {
    if (E > t->sort_cap || BN + 1 > t->sort_cap) {
      LB_HIP(hipStreamSynchronize(s));
      for (void* b : {(void*)t->snd_key, (void*)t->snd_perm, (void*)t->iota, (void*)t->snd_ptr, t->sort_tmp})
        if (b) (void)hipFree(b);
      t->snd_key = t->iota = nullptr;
      t->sort_tmp = nullptr;
      t->sort_tmp_bytes = 0;
      t->sort_cap = std::max<int64_t>(E + E / 8 + 1024, BN + 2);
      LB_HIP(hipMalloc((void**)&t->snd_perm, sizeof(int32_t) * t->sort_cap));
      LB_HIP(hipMalloc((void**)&t->snd_ptr, sizeof(int32_t) * t->sort_cap));
    }
    hipLaunchKernelGGL(k_sender_transpose, GRID1(std::max(E, BN + 1)), 0, s, e->row_ptr, e->senders, e->receivers, E, BN,
                       t->snd_perm, t->snd_ptr, t->dw_flag);
    return LB_OK;
  }
  if (E > t->sort_cap || BN + 1 > t->sort_cap || !t->sort_tmp) {
    LB_HIP(hipStreamSynchronize(s));
    for (void* b : {(void*)t->snd_key, (void*)t->snd_perm, (void*)t->iota, (void*)t->snd_ptr, t->sort_tmp})
      if (b) (void)hipFree(b);
    t->snd_key = t->snd_perm = t->iota = t->snd_ptr = nullptr;
    t->sort_tmp = nullptr;
    t->sort_cap = std::max<int64_t>(E + E / 8 + 1024, BN + 2);
    LB_HIP(hipMalloc((void**)&t->snd_key, sizeof(int32_t) * t->sort_cap));
    LB_HIP(hipMalloc((void**)&t->snd_perm, sizeof(int32_t) * t->sort_cap));
    LB_HIP(hipMalloc((void**)&t->iota, sizeof(int32_t) * t->sort_cap));
    LB_HIP(hipMalloc((void**)&t->snd_ptr, sizeof(int32_t) * t->sort_cap));
    hipLaunchKernelGGL(k_iota, GRID1(t->sort_cap), 0, s, t->iota, t->sort_cap);
    t->sort_tmp_bytes = 0;
    LB_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, t->sort_tmp_bytes, e->senders, t->snd_key, t->iota, t->snd_perm,
                                              (int)t->sort_cap, 0, 32, s));
    LB_HIP(hipMalloc(&t->sort_tmp, t->sort_tmp_bytes));
  }
  size_t bytes = t->sort_tmp_bytes;
  LB_HIP(hipcub::DeviceRadixSort::SortPairs(t->sort_tmp, bytes, e->senders, t->snd_key, t->iota, t->snd_perm, (int)E, 0, 32, s));
  hipLaunchKernelGGL(k_lower_bounds, GRID1(BN + 1), 0, s, t->snd_key, E, BN, t->snd_ptr);
  return LB_OK;
}

// ------------------------------------------------------------------------------------------------ C ABI
extern "C" int lb_gns_train_create(lb_engine* e, const lb_gns_desc* d, const float* w, int64_t n_floats,
                                   lb_gns_train** out) {
  if (!e || !d || !w || !out) return lb_fail(LB_ERR_ARG, "null argument");
  if (d->latent_size < 4 || d->latent_size > TD || d->blocks_per_step < 2 || d->blocks_per_step > 8)
    return lb_fail(LB_ERR_UNSUPPORTED, "training path: latent_size <= 128 and 2 <= num_mlp_layers <= 8 are built");
  if (d->out_dim != e->g.dim || d->node_in != e->g.node_in || d->edge_in != e->g.dim + 1)
    return lb_fail(LB_ERR_ARG, "model widths do not match the case");
  const int L = d->num_mp_steps;
  const bool has_emb = d->num_particle_types > 1;
  const int emb = has_emb ? d->embedding_size : 0;
  lb_gns_train* t = new lb_gns_train();
  t->desc = *d;
  t->eng = e;
  t->nlin = d->blocks_per_step;
  const int nmid = t->nlin - 2;
  t->nin = d->node_in + emb;
  t->kpad = (t->nin + 31) / 32 * 32;
  int64_t o = 0;
  t->off_embed = 0;
  if (has_emb) o += (int64_t)d->num_particle_types * emb;
  auto mlp = [&](int in, int outw, bool ln) {
    lb_train_mlp p{};
    p.in = in;
    p.out = outw;
    p.ln = ln;
    p.w0 = o; o += (int64_t)in * TD;
    p.b0 = o; o += TD;
    for (int m = 0; m < nmid; ++m) {   // GNS.flatten order: linear_0, linear_1, ..., LayerNorm
      p.wm.push_back(o); o += (int64_t)TD * TD;
      p.bm.push_back(o); o += TD;
    }
    p.w1 = o; o += (int64_t)TD * outw;
    p.b1 = o; o += outw;
    if (ln) {
      p.lns = o; o += outw;
      p.lno = o; o += outw;
    }
    return p;
  };
  t->enc_node = mlp(t->nin, TD, true);
  t->enc_edge = mlp(d->edge_in, TD, true);
  for (int k = 0; k < L; ++k) {
    t->pe.push_back(mlp(3 * TD, TD, true));
    t->pn.push_back(mlp(2 * TD, TD, true));
  }
  t->dec = mlp(TD, d->out_dim, false);
  // Latents narrower than 128 (round 4; GNS-5-64 of docs/pages/baselines.rst): the device blobs keep the 128-wide layout with
  // zeros in the padded rows / columns - they stay zero under the step (a padded hidden unit is relu(0) = 0, so its weight
  // gradients vanish; padded LayerNorm columns are masked in k_ln_bwd2; AdamW of (w, g, m, v) = 0 is 0) - and the caller's
  // blob (GNS.flatten with the model's latent size) is scattered into / gathered from it through an index map.
  const int lat = d->latent_size;
  t->lat = lat;
  int64_t oc = has_emb ? (int64_t)d->num_particle_types * emb : 0;
  if (lat != TD) t->cmap.reserve((size_t)o);
  for (int64_t i = 0; i < oc && lat != TD; ++i) t->cmap.push_back(i);  // embedding table: as is
  auto map_mlp = [&](const lb_train_mlp& p, int in_c, int nblk, int out_c) {
    // in_c caller rows of w0: nblk blocks of `lat` latent rows (row j of block q -> device row 128 q + j), or plain rows (nblk 0)
    if (lat != TD) {
      for (int r = 0; r < in_c; ++r) {
        const int rp = nblk ? (r / lat) * TD + (r % lat) : r;
        for (int c = 0; c < lat; ++c) t->cmap.push_back(p.w0 + (int64_t)rp * TD + c);
      }
      for (int c = 0; c < lat; ++c) t->cmap.push_back(p.b0 + c);
      for (size_t m = 0; m < p.wm.size(); ++m) {
        for (int r = 0; r < lat; ++r)
          for (int c = 0; c < lat; ++c) t->cmap.push_back(p.wm[m] + (int64_t)r * TD + c);
        for (int c = 0; c < lat; ++c) t->cmap.push_back(p.bm[m] + c);
      }
      for (int r = 0; r < lat; ++r)
        for (int c = 0; c < out_c; ++c) t->cmap.push_back(p.w1 + (int64_t)r * p.out + c);
      for (int c = 0; c < out_c; ++c) t->cmap.push_back(p.b1 + c);
      if (p.ln) {
        for (int c = 0; c < out_c; ++c) t->cmap.push_back(p.lns + c);
        for (int c = 0; c < out_c; ++c) t->cmap.push_back(p.lno + c);
      }
    }
    oc += (int64_t)in_c * lat + lat + (int64_t)p.wm.size() * ((int64_t)lat * lat + lat) + (int64_t)lat * out_c + out_c +
          (p.ln ? 2 * out_c : 0);
  };
  map_mlp(t->enc_node, t->nin, 0, lat);
  map_mlp(t->enc_edge, d->edge_in, 0, lat);
  for (int k = 0; k < L; ++k) {
    map_mlp(t->pe[k], 3 * lat, 3, lat);
    map_mlp(t->pn[k], 2 * lat, 2, lat);
  }
  map_mlp(t->dec, lat, 1, d->out_dim);
  if (oc != n_floats) {
    delete t;
    return lb_fail(LB_ERR_ARG, "weight blob has %lld floats, expected %lld", (long long)n_floats, (long long)oc);
  }
  t->n_floats = o;
  t->n_compact = oc;
  t->nlat.assign(L + 1, nullptr);
  t->elat.assign(L + 1, nullptr);
  t->ae.assign(L, nullptr); t->ze.assign(L, nullptr);
  t->xn.assign(L, nullptr); t->an.assign(L, nullptr); t->zn.assign(L, nullptr);
  int rc = LB_OK;
  for (float** p : {&t->w, &t->g, &t->m, &t->v})
    if (!rc) rc = lb_alloc(p, (size_t)o);
  if (!rc) rc = lb_alloc(&t->loss_dev, 1);
  if (!rc) rc = lb_alloc(&t->dw_flag, (size_t)(1 + LB_DW_CALLS));
  if (!rc && hipMemset(t->dw_flag, 0, sizeof(int32_t) * (1 + LB_DW_CALLS)) != hipSuccess) rc = lb_fail(LB_ERR_HIP, "hipMemset");
  if (!rc) rc = lb_alloc(&t->cnt_dev, (size_t)e->g.B);
  if (!rc) {
    std::vector<float> padded;
    const float* src = w;
    if (!t->cmap.empty()) {
      padded.assign((size_t)o, 0.f);
      for (int64_t i = 0; i < oc; ++i) padded[(size_t)t->cmap[(size_t)i]] = w[i];
      src = padded.data();
    }
    if (hipMemcpy(t->w, src, sizeof(float) * o, hipMemcpyHostToDevice) != hipSuccess ||
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

static void sgt_free(lb_gns_train* t);
extern "C" void lb_gns_train_destroy(lb_gns_train* t) {
  if (!t) return;
  sgt_free(t);
  std::vector<void*> bufs = {t->w, t->g, t->m, t->v, t->xnode, t->a_en, t->z_en, t->a_ee, t->z_ee, t->a_d, t->pred,
                             t->dn, t->de, t->dy, t->dz, t->da, t->dx, t->dagg, t->agg, t->dwpart, t->red_dev, t->proj, t->node_w,
                             t->loss_dev, t->dw_flag, t->da2, t->loss_part, t->cnt_dev, t->snd_key, t->snd_perm, t->iota, t->snd_ptr, t->sort_tmp,
                             t->wpack, t->pack_dev, t->pack_dev_h, t->wsc, t->tmax};
  for (auto* v : {&t->nlat, &t->elat, &t->ae, &t->ze, &t->xn, &t->an, &t->zn})
    for (float* p : *v) bufs.push_back(p);
  for (void* b : bufs)
    if (b) (void)hipFree(b);
  if (t->red_host) (void)hipHostFree(t->red_host);
  delete t;
}

// value_and_grad of _mse, summed over the batch (trainer.py:63-89), on the engine's CURRENT window / neighbor list.
// target_dev: (B*N, dim) fp32 normalised accelerations.  Gradients ACCUMULATE into the gradient blob (zero it with
// lb_gns_train_zero_grad); *loss_out = mean over the batch of the per-trajectory losses (host-synchronous).
static int gns_train_loss_grad_once(lb_gns_train* t, const float* target_dev, float loss_weight, double* loss_out,
                                    float* pred_out_dev);
static int segnn_train_loss_grad_once(lb_gns_train* t, const float* target_dev, float loss_weight, double* loss_out,
                                      float* pred_out_dev);
// One training step's loss + gradients with the step's guard flag around it: if a k_dw_part_h block saw activations outside
// [2^-8, 2^15) (bits 1 | 2), or k_sender_transpose an edge without its transpose (bit 4), the reductions have added nothing
// (k_part_reduce / k_embed_grad read the flag) and the step runs again - with X scaled per chunk from now on, or with
// the radix sort for this step.
// the step's loss and guard flag come back through pinned words behind the descriptor table (ONE stream synchronisation per
// step: round 5 took the loss and the flag with a pageable copy each)
static_assert(sizeof(lb_red_ent) >= sizeof(double) + sizeof(int32_t), "status words live in one spare descriptor");
static double* status_loss(lb_gns_train* t) { return reinterpret_cast<double*>(t->red_host + LB_RED_MAX); }
static int32_t* status_flag(lb_gns_train* t) { return reinterpret_cast<int32_t*>(status_loss(t) + 1); }
static int train_loss_grad_guarded(lb_gns_train* t, const float* target_dev, float loss_weight, double* loss_out,
                                   float* pred_out_dev) {
  auto once = [&]() {
    return t->sg ? segnn_train_loss_grad_once(t, target_dev, loss_weight, loss_out, pred_out_dev)
                 : gns_train_loss_grad_once(t, target_dev, loss_weight, loss_out, pred_out_dev);
  };
  hipStream_t s = t->eng->stream;
  int rc = LB_OK;
  std::vector<int32_t> slots;
  for (int attempt = 0; attempt < 4; ++attempt) {
    LB_HIP(hipMemsetAsync(t->dw_flag, 0, sizeof(int32_t) * (1 + LB_DW_CALLS), s));
    t->dw_call = 0;
    rc = once();
    if (rc) break;
    LB_HIP(hipMemcpyAsync(status_flag(t), t->dw_flag, sizeof(int32_t), hipMemcpyDeviceToHost, s));
    LB_HIP(hipStreamSynchronize(s));   // (also delivers the loss the attempt left in status_loss)
    const int32_t flag = *status_flag(t);
    if (!flag) break;
    // the reductions of this attempt added nothing (k_part_reduce / k_embed_grad saw the flag): repeat with what it asks for
    if (attempt == 3) { rc = lb_fail(LB_ERR_STATE, "training: the step's guard flag (%d) does not clear", flag); break; }
    if (flag & 4) {
      if (t->cub_sort) { rc = lb_fail(LB_ERR_STATE, "training: sender view failed with the radix sort"); break; }
      t->cub_sort = true;
    }
    if (flag & 3) {   // re-centre the X exponent of every call site that fired: largest |X| -> [2^3, 2^4)
      const int n = std::min(t->dw_call, LB_DW_CALLS);
      slots.assign((size_t)n, 0);
      LB_HIP(hipMemcpy(slots.data(), t->dw_flag + 1, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost));
      for (int c = 0; c < n; ++c) {
        if (!slots[c]) continue;
        const int ex = (int)(((uint32_t)slots[c] >> 23) & 0xffu);   // biased exponent of the call's largest raw |X|
        const int want = std::max(-100, std::min(100, 130 - ex));
        if (ex >= 255 || t->dw_xdyn[c]) continue;                    // (non-finite activations: non-finite gradients in any arithmetic)
        if (attempt > 0 || t->dw_xexp[c] == want) t->dw_xdyn[c] = 1;   // fired again after a re-centring: its chunks need their own scales
        t->dw_xexp[c] = (int8_t)want;
      }
      ++t->dw_fallbacks;
    }
  }
  t->cub_sort = false;
  if (!rc && loss_out) *loss_out = *status_loss(t);
  return rc;
}
extern "C" int32_t lb_gns_train_math_fallbacks(lb_gns_train* t) { return t ? t->dw_fallbacks : -1; }

extern "C" int lb_gns_train_loss_grad(lb_gns_train* t, const float* target_dev, float loss_weight, double* loss_out,
                                      float* pred_out_dev) {
  if (!t || !target_dev) return lb_fail(LB_ERR_ARG, "null argument");
  return train_loss_grad_guarded(t, target_dev, loss_weight, loss_out, pred_out_dev);
}
static int gns_train_loss_grad_once(lb_gns_train* t, const float* target_dev, float loss_weight, double* loss_out,
                                    float* pred_out_dev) {
  lb_engine* e = t->eng;
  if (e->e_cap <= 0) return lb_fail(LB_ERR_STATE, "lb_gns_train_loss_grad before lb_nl_allocate");
  hipStream_t s = e->stream;
  LB_HIP(hipMemcpyAsync(e->ctrl_host, e->ctrl, sizeof(lb_ctrl), hipMemcpyDeviceToHost, s));
  LB_HIP(hipStreamSynchronize(s));
  if (e->ctrl_host->overflow_step >= 0) return lb_fail(LB_ERR_STATE, "neighbor list overflowed: re-allocate first");
  const int64_t E = e->ctrl_host->n_edges_total, BN = e->BN;
  const int L = t->desc.num_mp_steps, dim = t->desc.out_dim;
  LB_TRY(train_ensure(t, BN, E));
  pack_all(t);  // the Linear operands in fragment order, from the current weights
  t->red_tab.clear();
  t->red_off = 0;
  t->red_blocks = 0;
  const bool has_emb = t->desc.num_particle_types > 1;
  const int emb = has_emb ? t->desc.embedding_size : 0;
  // ---- features (engine kernels): node row [features | embedding | 0-pad], edge features from the neighbor build
  const int kpad_saved = e->g.kpad;
  e->g.kpad = t->kpad;
  int rc = lbk_node_features(e, t->xnode, has_emb ? t->w + t->off_embed : nullptr, emb, t->desc.num_particle_types, nullptr,
                             nullptr, nullptr, nullptr);
  e->g.kpad = kpad_saved;
  if (rc) return rc;
  // ---- forward
  LB_TRY(mlp_fwd(t, t->enc_node, BN, t->xnode, t->kpad, t->a_en, t->z_en, nullptr, t->nlat[0], t->hs_n));
  LB_TRY(mlp_fwd(t, t->enc_edge, E, e->efeat, 8, t->a_ee, t->z_ee, nullptr, t->elat[0], t->hs_e));
  for (int k = 0; k < L; ++k) {
    // e' = LN(MLP([n_s | n_r | e])) is both the message and (plus e) the next edge latent: keep e' in dy, then residual
    LB_TRY(edge_fwd(t, t->pe[k], E, BN, t->nlat[k], t->elat[k], t->ae[k], t->ze[k], t->dy, t->elat[k + 1]));
    hipLaunchKernelGGL(k_seg_sum, GRID1(BN * 32), 0, s, e->row_ptr, t->dy, t->agg, BN, E, t->nlat[k], t->xn[k]);
    LB_TRY(mlp_fwd(t, t->pn[k], BN, t->xn[k], 2 * TD, t->an[k], t->zn[k], t->nlat[k], t->nlat[k + 1], t->hs_n));
  }
  LB_TRY(mlp_fwd(t, t->dec, BN, t->nlat[L], TD, t->a_d, nullptr, nullptr, t->pred, t->hs_n));
  if (pred_out_dev) LB_HIP(hipMemcpyAsync(pred_out_dev, t->pred, sizeof(float) * BN * dim, hipMemcpyDeviceToDevice, s));
  // ---- loss and d loss / d pred, the sender-sorted view of this step's edge list
  LB_TRY(train_loss(t, t->pred, target_dev, loss_weight, t->dy));
  LB_TRY(train_sender_sort(t, E, BN));
  // ---- backward
  LB_TRY(mlp_bwd(t, t->dec, BN, t->nlat[L], TD, t->a_d, nullptr, t->dy, t->dn, t->hs_n));  // dn = d loss / d n_L
  LB_HIP(hipMemsetAsync(t->de, 0, sizeof(float) * std::max<int64_t>(E, 1) * TD, s));  // e_L has no reader
  for (int k = L - 1; k >= 0; --k) {
    // node block: n_{k+1} = n_k + LN(MLP([n_k | agg_k])): dy = dn (also flows to n_k through the residual)
    // d [n_k | agg_k] = da W0^T lands where it is used: the n_k half is added to dn, the agg_k half is dagg (round 4 wrote
    // the 256-wide product and split it in a pass of its own).  dn was consumed (LayerNorm backward) before it is updated.
    LB_TRY(mlp_bwd(t, t->pn[k], BN, t->xn[k], 2 * TD, t->an[k], t->zn[k], t->dn, nullptr, t->hs_n));
    LB_TRY(gemm_nt_pair(t, BN, t->da, t->w + t->pn[k].w0, t->dn, 1.f, t->w + t->pn[k].w0 + (size_t)TD * TD, t->dagg, 0.f));
    LB_TRY(edge_bwd(t, t->pe[k], E, BN, t->nlat[k], t->elat[k], t->ae[k], t->ze[k], t->dagg, t->de, t->dn));
  }
  LB_TRY(mlp_bwd(t, t->enc_edge, E, e->efeat, 8, t->a_ee, t->z_ee, t->de, nullptr, t->hs_e));
  LB_TRY(mlp_bwd(t, t->enc_node, BN, t->xnode, t->kpad, t->a_en, t->z_en, t->dn, has_emb ? t->dx : nullptr, t->hs_n));
  if (has_emb)
    hipLaunchKernelGGL(k_embed_grad, dim3(t->desc.num_particle_types), dim3(1024), 0, s, t->dx, t->kpad, t->desc.node_in, emb,
                       e->ptype, t->desc.num_particle_types, BN, t->g + t->off_embed, t->dw_flag);
  LB_TRY(red_flush(t));  // every weight / bias / LayerNorm gradient: partials -> gradient blob, one launch
  LB_HIP(hipGetLastError());
  if (loss_out)   // (train_loss_grad_guarded synchronises and hands it over)
    LB_HIP(hipMemcpyAsync(status_loss(t), t->loss_dev, sizeof(double), hipMemcpyDeviceToHost, s));
  return LB_OK;
}

extern "C" int lb_gns_train_zero_grad(lb_gns_train* t) {
  if (!t) return lb_fail(LB_ERR_ARG, "null argument");
  LB_HIP(hipMemsetAsync(t->g, 0, sizeof(float) * t->n_floats, t->eng->stream));
  return LB_OK;
}

// optax.adamw step on every parameter (trainer.py:189-193, :86-87); the step counter of the bias correction lives
// in the handle (set_step restores it from a checkpoint)
extern "C" int lb_adamw_step(lb_gns_train* t, float lr, float b1, float b2, float eps, float weight_decay) {
  if (!t) return lb_fail(LB_ERR_ARG, "null argument");
  t->step += 1;
  const float c1 = 1.f - powf(b1, (float)t->step), c2 = 1.f - powf(b2, (float)t->step);
  hipLaunchKernelGGL(k_adamw, GRID1(t->n_floats), 0, t->eng->stream, t->w, t->g, t->m, t->v, t->n_floats, lr, b1, b2, eps,
                     weight_decay, c1, c2);
  LB_HIP(hipGetLastError());
  return LB_OK;
}

// which: 0 weights, 1 gradients, 2 first moment, 3 second moment (flat blob, GNS.flatten order)
extern "C" int lb_gns_train_read(lb_gns_train* t, int32_t which, float* out_host, int64_t n_floats) {
  if (!t || !out_host || n_floats != t->n_compact || which < 0 || which > 3) return lb_fail(LB_ERR_ARG, "bad argument");
  const float* src = which == 0 ? t->w : which == 1 ? t->g : which == 2 ? t->m : t->v;
  LB_HIP(hipStreamSynchronize(t->eng->stream));
  if (t->cmap.empty()) {
    LB_HIP(hipMemcpy(out_host, src, sizeof(float) * n_floats, hipMemcpyDeviceToHost));
  } else {
    std::vector<float> padded((size_t)t->n_floats);
    LB_HIP(hipMemcpy(padded.data(), src, sizeof(float) * t->n_floats, hipMemcpyDeviceToHost));
    for (int64_t i = 0; i < n_floats; ++i) out_host[i] = padded[(size_t)t->cmap[(size_t)i]];
  }
  return LB_OK;
}
extern "C" int64_t lb_gns_train_step_count(lb_gns_train* t) { return t ? (int64_t)t->step : -1; }

extern "C" int lb_gns_train_write(lb_gns_train* t, int32_t which, const float* in_host, int64_t n_floats, int64_t step) {
  if (!t || !in_host || n_floats != t->n_compact || which < 0 || which > 3) return lb_fail(LB_ERR_ARG, "bad argument");
  float* dst = which == 0 ? t->w : which == 1 ? t->g : which == 2 ? t->m : t->v;
  LB_HIP(hipStreamSynchronize(t->eng->stream));
  if (t->cmap.empty()) {
    LB_HIP(hipMemcpy(dst, in_host, sizeof(float) * n_floats, hipMemcpyHostToDevice));
  } else {
    std::vector<float> padded((size_t)t->n_floats, 0.f);
    for (int64_t i = 0; i < n_floats; ++i) padded[(size_t)t->cmap[(size_t)i]] = in_host[i];
    LB_HIP(hipMemcpy(dst, padded.data(), sizeof(float) * t->n_floats, hipMemcpyHostToDevice));
  }
  if (step >= 0) t->step = step;
  return LB_OK;
}

// ------------------------------------------------------------------------------------------------ SEGNN
#include "lb_train_segnn.h"
