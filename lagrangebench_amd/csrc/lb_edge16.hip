// lb_edge16.hip - the edge MLP on 16-row tiles (v_mfma_f32_16x16x4_f32), software-pipelined.
//
// Same mathematics and the same "register-chained transposed MLP" idea as k_edge_mlp in lb_gns.hip
// (reference: GNS._processor update_edge_features + jraph segment_sum, models/gns.py:86-122), but
// the per-wave tile is 16 edges instead of 32.  Halving the tile halves every per-wave register
// array (accumulators 32, activations 32), which buys room - at two waves per SIMD - for
//   * keeping the edge latents `e` resident for the residual (no second read),
//   * a full software prefetch of the NEXT tile's inputs (e row, gathered sender/receiver
//     projections) and of the tile after that's indices, issued before the current tile's GEMMs,
// so global-memory latency and the epilogue of one wave hide behind the other wave's MFMAs.
//
// Layouts (lane l: n = l & 15 is the edge inside the tile, g = l >> 4 the feature quarter):
//   C/D of 16x16x4:  acc[mb][j]  <->  feature 16*mb + 4*g + j of edge n      (mb = 0..7, j = 0..3)
//   B operand:       lane (n,g) supplies X[n][k_g]; the MFMA "reg j of block mbk" contracts the four
//                    features 16*mbk + 4*g' + j, g' = 0..3  ->  the accumulator registers of one
//                    layer ARE the B operands of the next (K order is free).
//   A operand:       lane (m = l&15, g) supplies W[16*mbk + 4*g + j][16*mbo + m]; packed on the host as
//                    [(mbk*4 + j)][hf][lane][c] with mbo = 4*hf + c: two ds_read_b128 feed 8 MFMAs.
//   row-major rows:  a 128-float row is 32 16-byte chunks; lane (n,g) owns chunks 4*mb + g.
//   edge latents:    `elat` is private to the edge kernels, so it is stored TILE-BLOCKED,
//                    elat4[(tile*8 + mb)*64 + lane] = features 16*mb + 4*g .. +3 of edge 16*tile + n:
//                    every wave-level load/store of the latents is one contiguous 1 KiB (8 full
//                    128-B lines) instead of 16 half-used lines of 16 different rows.
// The fused aggregation works on rows of 16 lanes = exactly one DPP row (row_shr 1,2,4,8).
#include <stdlib.h>
#include <string.h>

#include "lb_device.h"

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// ---- fp16x2 split mode ("f16x2"): every fp32 operand x is carried as hi = fp16(x), lo = fp16(x - hi)
// (|x - hi - lo| <= 2^-24 |x| in the fp16 normal range), and a product is evaluated as
// lo_a*hi_b + hi_a*lo_b + hi_a*hi_b on v_mfma_f32_16x16x32_f16 - each partial product is exact in
// the fp32 accumulator, the dropped lo*lo term is <= 2^-24 |ab|.  Three MFMAs at the 2.5 PF fp16
// rate instead of one at the 157 TF fp32 rate: ~5x fewer matrix-pipe cycles for fp32-class
// accuracy (measured: 10-layer GNS output within 1.2e-6 of the fp32-MFMA path).
// K-step of 32: lane (n, g) supplies its 8 features {32p + 4g + i, 32p + 16 + 4g + i} (i<4), i.e.
// the accumulator registers of blocks 2p and 2p+1 - the chained-layer property is kept.
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
#define MFMA16H(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16((a), (b), (c), 0, 0, 0)

static uint16_t lb_f32_to_f16_rne(float f) {
  uint32_t x;
  memcpy(&x, &f, 4);
  const uint32_t sign = (x >> 16) & 0x8000u;
  x &= 0x7fffffffu;
  if (x >= 0x47800000u) return (uint16_t)(sign | (x > 0x7f800000u ? 0x7e00u : 0x7c00u));  // inf / nan
  if (x < 0x38800000u) {  // subnormal half (or zero): value * 2^24 rounded to nearest even
    if (x < 0x33000000u) return (uint16_t)sign;
    const int shift = 126 - (int)(x >> 23);  // 14 .. 24
    uint32_t m = (x & 0x7fffffu) | 0x800000u;
    const uint32_t lsb = 1u << shift, half = lsb >> 1;
    uint32_t r = m >> shift;
    const uint32_t rem = m & (lsb - 1);
    if (rem > half || (rem == half && (r & 1))) ++r;
    return (uint16_t)(sign | r);
  }
  uint32_t r = x - 0x38000000u;  // rebias exponent 127 -> 15
  const uint32_t rem = r & 0x1fffu;
  r >>= 13;
  if (rem > 0x1000u || (rem == 0x1000u && (r & 1))) ++r;
  return (uint16_t)(sign | r);
}
static float lb_f16_to_f32(uint16_t h) {
  const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
  const uint32_t e = (h >> 10) & 0x1f, m = h & 0x3ffu;
  uint32_t x;
  if (e == 0) {
    if (m == 0) {
      x = sign;
    } else {
      int k = 0;
      uint32_t mm = m;
      while (!(mm & 0x400u)) {
        mm <<= 1;
        ++k;
      }
      x = sign | ((uint32_t)(113 - k) << 23) | ((mm & 0x3ffu) << 13);
    }
  } else if (e == 31) {
    x = sign | 0x7f800000u | (m << 13);
  } else {
    x = sign | ((e + 112) << 23) | (m << 13);
  }
  float f;
  memcpy(&f, &x, 4);
  return f;
}

// out: Kpad*Mpad "floats" worth of storage holding [(p*NMBO + mbo)][part: 0 hi, 1 lo][lane][8 halfs]
void lb_pack_weight16h(const float* w, int K, int M, int Kpad, float* out, int Mpad) {
  uint16_t* o = reinterpret_cast<uint16_t*>(out);
  const int NP = Kpad / 32, NMBO = Mpad / 16;
  for (int p = 0; p < NP; ++p)
    for (int mbo = 0; mbo < NMBO; ++mbo)
      for (int lane = 0; lane < 64; ++lane)
        for (int i = 0; i < 8; ++i) {
          const int g = lane >> 4;
          const int k = 32 * p + (i < 4 ? 4 * g + i : 16 + 4 * g + (i - 4));
          const int m = 16 * mbo + (lane & 15);
          const float x = (k < K && m < M) ? w[(size_t)k * M + m] : 0.f;
          const uint16_t hi = lb_f32_to_f16_rne(x);
          const uint16_t lo = lb_f32_to_f16_rne(x - lb_f16_to_f32(hi));
          const size_t base = ((size_t)(p * NMBO + mbo) * 2) * 64;
          o[((base + lane) * 8) + i] = hi;
          o[((base + 64 + lane) * 8) + i] = lo;
        }
}

typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
typedef _Float16 h4_t __attribute__((ext_vector_type(4)));
// x -> hi = fp16(x) (RNE), lo = fp16(x - hi): written on 2-vectors so that hipcc emits the packed
// v_cvt_pk_f16_f32 / v_pk_add_f32 forms (5 VALU per two elements instead of ~10).
__device__ __forceinline__ void lb_split8(const f32x4& x0, const f32x4& x1, h8& hi, h8& lo) {
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

// acc[0..7] += W^T * B over NP blocks of 32 k's in f16x2 arithmetic.  ld(p, mbo, part) -> 16 B.
template <int NP, typename LD>
__device__ __forceinline__ void lb_gemm16h(LD ld, const f32x4 (&v)[2 * NP], f32x4 (&acc)[8]) {
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    h8 bh, bl;
    lb_split8(v[2 * p], v[2 * p + 1], bh, bl);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      h8 ah[4], al[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        ah[c] = __builtin_bit_cast(h8, ld(p, 4 * q + c, 0));
        al[c] = __builtin_bit_cast(h8, ld(p, 4 * q + c, 1));
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[4 * q + c] = MFMA16H(al[c], bh, acc[4 * q + c]);
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[4 * q + c] = MFMA16H(ah[c], bl, acc[4 * q + c]);
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[4 * q + c] = MFMA16H(ah[c], bh, acc[4 * q + c]);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}
#define E16_THREADS 512
#define E16_WAVES 8

void lb_pack_weight16(const float* w, int K, int M, int Kpad, float* out) {
  // M <= 128 (8 output blocks of 16); Kpad multiple of 16; out has Kpad*128 floats.
  const int NMBK = Kpad / 16;
  for (int mbk = 0; mbk < NMBK; ++mbk)
    for (int j = 0; j < 4; ++j)
      for (int hf = 0; hf < 2; ++hf)
        for (int lane = 0; lane < 64; ++lane)
          for (int c = 0; c < 4; ++c) {
            const int k = 16 * mbk + 4 * (lane >> 4) + j;
            const int m = 16 * (4 * hf + c) + (lane & 15);
            out[((((size_t)(mbk * 4 + j) * 2 + hf) * 64 + lane) * 4) + c] =
                (k < K && m < M) ? w[(size_t)k * M + m] : 0.f;
          }
}

// acc[0..7] += W^T * B over NMBK blocks of 16 k's.  One step (mbk, j) = 2 ds_read_b128 + 8 MFMAs;
// the next step's fragments are fetched before the current step's MFMAs issue.
template <int NMBK, typename LD>
__device__ __forceinline__ void lb_gemm16(LD ld, const f32x4 (&v)[NMBK], f32x4 (&acc)[8]) {
  f32x4 a0 = ld(0, 0), a1 = ld(0, 1);
#pragma unroll
  for (int mbk = 0; mbk < NMBK; ++mbk) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int step = mbk * 4 + j;
      f32x4 n0 = a0, n1 = a1;
      if (step + 1 < NMBK * 4) {
        n0 = ld(step + 1, 0);
        n1 = ld(step + 1, 1);
      }
      const float b = v[mbk][j];
      acc[0] = MFMA16(a0[0], b, acc[0]);
      acc[1] = MFMA16(a0[1], b, acc[1]);
      acc[2] = MFMA16(a0[2], b, acc[2]);
      acc[3] = MFMA16(a0[3], b, acc[3]);
      acc[4] = MFMA16(a1[0], b, acc[4]);
      acc[5] = MFMA16(a1[1], b, acc[5]);
      acc[6] = MFMA16(a1[2], b, acc[6]);
      acc[7] = MFMA16(a1[3], b, acc[7]);
      a0 = n0;
      a1 = n1;
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

// ABL (ablation harnesses only, 0 in the product): 1 no Ps/Pr gather, 2 no e load, 4 no stores,
// 8 no LayerNorm, 16 no GEMM2, 32 no GEMM1, 64 no segmented scan.
template <bool PROC, bool F16, int ABL = 0>
__global__ void __launch_bounds__(E16_THREADS, 2) k_edge16(lb_edge16_args a) {
  // LDS: packed W0 (PROC: 128x128 edge rows of the first layer; ENC: 16x128, or 32x128 hi|lo in
  // f16x2 mode) then packed W1 (64 KiB in either mode: fp32, or fp16 hi + fp16 lo).
  constexpr int NW0 = PROC ? 4096 : (F16 ? 1024 : 512);
  // ... then the per-feature vectors b1 | ln_scale | ln_offset | b0 (32 f32x4 each): fetching them
  // from global memory in the epilogue would queue behind the prefetch loads (vmcnt is in-order).
  __shared__ f32x4 sW[NW0 + 4096 + 128];
  if (a.ctrl->overflow_step >= 0) return;
  const int tid = threadIdx.x;
  {
    const f32x4* g0 = reinterpret_cast<const f32x4*>(a.w0p);
    const f32x4* g1 = reinterpret_cast<const f32x4*>(a.w1p);
    for (int i = tid; i < NW0; i += E16_THREADS) sW[i] = g0[i];
    for (int i = tid; i < 4096; i += E16_THREADS) sW[NW0 + i] = g1[i];
    if (tid < 128) {
      const float* src = tid < 32 ? a.b1 : (tid < 64 ? a.ln_s : (tid < 96 ? a.ln_o : a.b0));
      sW[NW0 + 4096 + tid] = src ? reinterpret_cast<const f32x4*>(src)[tid & 31] : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  __syncthreads();
  const int E = a.ctrl->n_edges_total;
  const int ntiles = (E + 15) >> 4;
  const int lane = tid & 63, wave = tid >> 6;
  const int n = lane & 15, g = lane >> 4;
  // XCD-aware walk: block b runs on XCD b % 8; each XCD owns a contiguous eighth of the list.
  const int xcd = blockIdx.x & 7, slot = (blockIdx.x >> 3) * E16_WAVES + wave;
  const int stride = (gridDim.x >> 3) * E16_WAVES;
  const int t_lo = (int)(((int64_t)ntiles * xcd) >> 3), t_hi = (int)(((int64_t)ntiles * (xcd + 1)) >> 3);
  int t = t_lo + slot;
  if (t >= t_hi) return;

  auto ld0 = [&](int step, int hf) -> f32x4 { return sW[(step * 2 + hf) * 64 + lane]; };
  auto ld1 = [&](int step, int hf) -> f32x4 { return sW[NW0 + (step * 2 + hf) * 64 + lane]; };
  auto ldh0 = [&](int p, int mbo, int part) -> f32x4 { return sW[((p * 8 + mbo) * 2 + part) * 64 + lane]; };
  auto ldh1 = [&](int p, int mbo, int part) -> f32x4 {
    return sW[NW0 + ((p * 8 + mbo) * 2 + part) * 64 + lane];
  };
  auto rowc_of = [&](int tt) -> int64_t {
    const int row = tt * 16 + n;
    return row < E ? row : E - 1;
  };
  const f32x4* psr4 = reinterpret_cast<const f32x4*>(a.psr);
  const f32x4* b1_4 = &sW[NW0 + 4096];
  const f32x4* lns4 = &sW[NW0 + 4096 + 32];
  const f32x4* lno4 = &sW[NW0 + 4096 + 64];
  const f32x4* b0_4 = &sW[NW0 + 4096 + 96];

  // ---- software pipeline state: data of the tile about to be computed (issued one tile ago) and
  // the indices of the tile after it (issued two tiles ago).  EVERY load below is unconditional
  // (tile indices are clamped to the wave's last tile instead of being branched around): a load
  // under a branch makes hipcc merge the loop-carried registers through copies and wait
  // vmcnt(0) right after issuing them, which serialises the whole prefetch.
  const int n_iter = (t_hi - 1 - t) / stride + 1;
  const int t_last = t + (n_iter - 1) * stride;
  f32x4 ve_n[8], ps_n[8], pr_n[8];
  f32x4 vin_n = {0.f, 0.f, 0.f, 0.f};
  int s_n = 0, r_n = 0, r_pref = 0;
  auto issue = [&](int tt, int s, int r) {
    const int64_t rc = rowc_of(tt);
    if (PROC) {
      const f32x4* er = reinterpret_cast<const f32x4*>(a.elat) + (int64_t)tt * 512 + lane;
      const f32x4* ps = psr4 + (int64_t)s * 64 + g;
      const f32x4* pr = psr4 + (int64_t)r * 64 + 32 + g;
#pragma unroll
      for (int mb = 0; mb < 8; ++mb) {
        ve_n[mb] = (ABL & 2) ? f32x4{1.f, 2.f, (float)rc, (float)mb} : er[64 * mb];
        ps_n[mb] = (ABL & 1) ? f32x4{.1f, .2f, (float)s, (float)mb} : ps[4 * mb];
        pr_n[mb] = (ABL & 1) ? f32x4{.3f, .1f, (float)r, (float)mb} : pr[4 * mb];
      }
    } else {
      vin_n = reinterpret_cast<const f32x4*>(a.efeat)[rc * 2 + (g & 1)];
    }
  };
  {
    const int64_t rc = rowc_of(t);
    int s0 = 0, r0 = 0;
    if (PROC) {
      s0 = a.senders[rc];
      r0 = a.receivers[rc];
    }
    issue(t, s0, r0);
    r_pref = r0;
    if (PROC) {
      const int64_t rn = rowc_of(min(t + stride, t_last));
      s_n = a.senders[rn];
      r_n = a.receivers[rn];
    }
  }

  for (int it = 0; it < n_iter; ++it, t += stride) {
    // ---- take delivery of the prefetched tile
    f32x4 acc[8], ve[8];
    f32x4 vin[1];
    const int r_cur = r_pref;
    if (PROC) {
#pragma unroll
      for (int mb = 0; mb < 8; ++mb) {
        ve[mb] = ve_n[mb];
        acc[mb] = ps_n[mb] + pr_n[mb];
      }
    } else {
      vin[0] = (g < 2) ? vin_n : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int mb = 0; mb < 8; ++mb) acc[mb] = b0_4[4 * mb + g];
    }
    // ---- put the next tile's loads (and the one after's indices) in flight before the GEMMs
    // (the last iteration re-fetches its own tile: harmless, keeps the loads branch-free)
    issue(min(t + stride, t_last), s_n, r_n);
    r_pref = r_n;
    if (PROC) {
      const int64_t rn = rowc_of(min(t + 2 * stride, t_last));
      s_n = a.senders[rn];
      r_n = a.receivers[rn];
    }
    // ---- Linear -> ReLU -> Linear -> LayerNorm, all in registers
    if constexpr (ABL & 32) {
    } else if constexpr (F16) {
      if constexpr (PROC) {
        lb_gemm16h<4>(ldh0, ve, acc);
      } else {
        f32x4 vin2[2] = {vin[0], f32x4{0.f, 0.f, 0.f, 0.f}};
        lb_gemm16h<1>(ldh0, vin2, acc);
      }
    } else {
      if (PROC)
        lb_gemm16<8>(ld0, ve, acc);
      else
        lb_gemm16<1>(ld0, vin, acc);
    }
#pragma unroll
    for (int mb = 0; mb < 8; ++mb)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[mb][j] = fmaxf(acc[mb][j], 0.f);
    f32x4 acc2[8];
#pragma unroll
    for (int mb = 0; mb < 8; ++mb) acc2[mb] = b1_4[4 * mb + g];
    if constexpr (ABL & 16) {
#pragma unroll
      for (int mb = 0; mb < 8; ++mb) acc2[mb] = acc2[mb] + acc[mb];
    } else if constexpr (F16)
      lb_gemm16h<4>(ldh1, acc, acc2);
    else
      lb_gemm16<8>(ld1, acc, acc2);
    float sm = 0.f;
#pragma unroll
    for (int mb = 0; mb < 8; ++mb) sm += (acc2[mb][0] + acc2[mb][1]) + (acc2[mb][2] + acc2[mb][3]);
    sm += __shfl_xor(sm, 16);
    sm += __shfl_xor(sm, 32);
    const float mean = sm * a.ctrl->ln_inv_d;
    float vs = 0.f;
#pragma unroll
    for (int mb = 0; mb < 8; ++mb)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float d = acc2[mb][j] - mean;
        vs += d * d;
      }
    vs += __shfl_xor(vs, 16);
    vs += __shfl_xor(vs, 32);
    const float rs = 1.0f / sqrtf(fmaxf(vs - a.ctrl->ln_pad * (mean * mean), 0.f) * a.ctrl->ln_inv_d + 1e-5f);
    f32x4 y[8];
#pragma unroll
    for (int mb = 0; mb < 8; ++mb) {
      const f32x4 sc = lns4[4 * mb + g], of = lno4[4 * mb + g];
#pragma unroll
      for (int j = 0; j < 4; ++j)
        y[mb][j] = (ABL & 8) ? acc2[mb][j] : (sc[j] * rs) * (acc2[mb][j] - mean) + of[j];
    }
    // ---- stores
    const int row = t * 16 + n;
    const bool valid = row < E;
    const bool do_store = !((ABL & 4) && a.row_ptr[0] != -12345);
    if (do_store) {  // whole tile (rows past E are padding inside the blocked allocation)
      f32x4* er = reinterpret_cast<f32x4*>(a.elat) + (int64_t)t * 512 + lane;
      if (PROC) {
        // residual, gns.py:120-122; the updated latents of the LAST layer are never read again
        // (the decoder only takes the node latents, gns.py:125-133): their store is skipped
        if (!a.skip_elat_store) {
#pragma unroll
          for (int mb = 0; mb < 8; ++mb) er[64 * mb] = ve[mb] + y[mb];
        }
        if (!a.fused && valid) {
          f32x4* mr = reinterpret_cast<f32x4*>(a.msg) + (int64_t)row * 32 + g;
#pragma unroll
          for (int mb = 0; mb < 8; ++mb) mr[4 * mb] = y[mb];
        }
      } else {
#pragma unroll
        for (int mb = 0; mb < 8; ++mb) er[64 * mb] = y[mb];
      }
    }
    if (PROC && a.fused && !(ABL & 64)) {
      // fused jraph.segment_sum: segmented Hillis-Steele scan inside each 16-lane DPP row
      const int rr = valid ? r_cur : (-1 - n);
      const int r_prev = __builtin_amdgcn_update_dpp(-2, rr, 0x111, 0xF, 0xF, false);
      const bool head = (n == 0) || (rr != r_prev);
      const unsigned H = (unsigned)(__ballot(head) & 0xffffull);
      const unsigned below = H & ((2u << n) - 1u);
      const int segstart = 31 - __clz(below);
      const bool tail = (n == 15) || ((H >> (n + 1)) & 1u);
      // masks as 0/1 floats: x += shifted * m is ONE v_fmac_f32 with a DPP source per register and
      // step (a select would be three instructions); lanes past E contribute exact zeros
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
      if (tail && valid && do_store) {
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
}

// The fall-back launcher: exact fp32 (LB_MATH=f32 / the range guard's switch), the encoder in fp32, and the f16x2
// processor with STAND-ALONE aggregation (messages written for k_segment_sum).  The fused f16x2 processor runs on
// lb_edge16v.hip / lb_msplit.hip; round 1's three-wave k_edge16n is gone from the tree (round 4; profiles/r01_* have its numbers).
int lbk_edge16(lb_engine* e, const lb_edge16_args& a, bool proc, bool f16x2) {
  if (proc && f16x2)
    hipLaunchKernelGGL((k_edge16<true, true>), dim3(256), dim3(E16_THREADS), 0, e->stream, a);
  else if (proc)
    hipLaunchKernelGGL((k_edge16<true, false>), dim3(256), dim3(E16_THREADS), 0, e->stream, a);
  else
    hipLaunchKernelGGL((k_edge16<false, false>), dim3(256), dim3(E16_THREADS), 0, e->stream, a);
  LB_HIP(hipGetLastError());
  return LB_OK;
}
