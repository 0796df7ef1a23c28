// lb_node16h.hip - node MLP (+ residual + next-layer sender/receiver projection) in f16x2
// arithmetic, weights streamed through LDS.
//
// Reference: GNS._encoder node branch (models/gns.py:65-72) and the processor's
// update_node_features + residual (gns.py:103-113,120-122), plus the per-node half of the next
// edge MLP's first Linear (W0[:2D] split, see lb_gns.hip header).
//
// Why a different structure from the edge kernel: one node tile needs THREE weight matrices
// (W0 256x128, W1 128x128, projection 128x256 = 320 KiB as fp16 hi|lo), which do not fit the
// 160 KiB LDS, and in f16x2 arithmetic the matrix pipe consumes weight fragments far faster than
// L2 can deliver them to a single wave.  So a 512-thread workgroup (8 waves = 8 tiles of 16 nodes)
// walks the weights in 32 KiB chunks through a ring of three LDS buffers: chunks c+1 and c+2 are in
// flight (global loads into 2 x 16 staging VGPRs, committed to LDS one chunk ahead of use) while
// all 8 waves run their MFMAs on chunk c; one barrier per chunk.  Each weight byte is fetched from
// L2 once per 128 nodes instead of once per 32, and the L2 latency (~2 us) is covered by two chunks
// of MFMA work instead of one.
// Inside a wave the layers are chained in registers exactly as in lb_edge16.hip (16x16x32 fp16
// MFMA, C-layout lane (n = l&15, g = l>>4) holds features 16*mb + 4*g + j).
#include "lb_device.h"

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
#define MFMA16H(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16((a), (b), (c), 0, 0, 0)
#define N16_THREADS 256
#define N16_WAVES (N16_THREADS / 64)
#define N16_STG (2048 / N16_THREADS)  // f32x4 staged per thread and chunk
#define CHUNK_VEC 2048  // f32x4 per chunk (32 KiB)

typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
typedef _Float16 h4_t __attribute__((ext_vector_type(4)));
// x -> hi = fp16(x) (RNE), lo = fp16(x - hi): written on 2-vectors so that hipcc emits the packed
// v_cvt_pk_f16_f32 / v_pk_add_f32 forms (5 VALU per two elements instead of ~10).
__device__ __forceinline__ void lb_split8n(const f32x4& x0, const f32x4& x1, h8& hi, h8& lo) {
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

// 4 output blocks (mbo0..mbo0+3) of one k-step p from the LDS chunk: frag index inside the chunk
// is ((pp*NMBO + mbo)*2 + part)*64 + lane.
struct lb_frag4 {
  h8 ah[4], al[4];
};
template <int NMBO>
__device__ __forceinline__ void lb_quad_load(const f32x4* __restrict__ buf, int pp, int mbo0, int lane,
                                             lb_frag4& f) {
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    f.ah[c] = __builtin_bit_cast(h8, buf[((pp * NMBO + mbo0 + c) * 2 + 0) * 64 + lane]);
    f.al[c] = __builtin_bit_cast(h8, buf[((pp * NMBO + mbo0 + c) * 2 + 1) * 64 + lane]);
  }
}
// three passes over 4 independent accumulators (the spacing the f16 MFMA accumulate chain needs)
__device__ __forceinline__ void lb_quad_mfma(const lb_frag4& f, const h8& bh, const h8& bl, f32x4* acc) {
#pragma unroll
  for (int c = 0; c < 4; ++c) acc[c] = MFMA16H(f.al[c], bh, acc[c]);
#pragma unroll
  for (int c = 0; c < 4; ++c) acc[c] = MFMA16H(f.ah[c], bl, acc[c]);
#pragma unroll
  for (int c = 0; c < 4; ++c) acc[c] = MFMA16H(f.ah[c], bh, acc[c]);
}
// One LDS chunk = 4 quads.  The fragments of quad i+1 are fetched (8 ds_read_b128) before the 12
// MFMAs of quad i issue, so LDS latency hides behind the matrix pipe instead of adding to it.
// GEMM chunk (NMBO = 8): quads (pp 0, blocks 0-3), (pp 0, 4-7), (pp 1, 0-3), (pp 1, 4-7), B operand
// b0 for pp 0 and b1 for pp 1, n_pp = number of valid k-steps in the chunk.
__device__ __forceinline__ void lb_chunk_gemm8(const f32x4* __restrict__ buf, int lane, int n_pp,
                                               const h8& b0h, const h8& b0l, const h8& b1h,
                                               const h8& b1l, f32x4* acc) {
  lb_frag4 f0, f1;
  lb_quad_load<8>(buf, 0, 0, lane, f0);
  __builtin_amdgcn_sched_barrier(0);
  lb_quad_load<8>(buf, 0, 4, lane, f1);
  lb_quad_mfma(f0, b0h, b0l, &acc[0]);
  __builtin_amdgcn_sched_barrier(0);
  if (n_pp > 1) lb_quad_load<8>(buf, 1, 0, lane, f0);
  lb_quad_mfma(f1, b0h, b0l, &acc[4]);
  __builtin_amdgcn_sched_barrier(0);
  if (n_pp > 1) {
    lb_quad_load<8>(buf, 1, 4, lane, f1);
    lb_quad_mfma(f0, b1h, b1l, &acc[0]);
    __builtin_amdgcn_sched_barrier(0);
    lb_quad_mfma(f1, b1h, b1l, &acc[4]);
    __builtin_amdgcn_sched_barrier(0);
  }
}
// projection chunk (NMBO = 16, one k-step): quads over blocks 0-3, 4-7, 8-11, 12-15
__device__ __forceinline__ void lb_chunk_proj16(const f32x4* __restrict__ buf, int lane, const h8& bh,
                                                const h8& bl, f32x4* acc) {
  lb_frag4 f0, f1;
  lb_quad_load<16>(buf, 0, 0, lane, f0);
  __builtin_amdgcn_sched_barrier(0);
  lb_quad_load<16>(buf, 0, 4, lane, f1);
  lb_quad_mfma(f0, bh, bl, &acc[0]);
  __builtin_amdgcn_sched_barrier(0);
  lb_quad_load<16>(buf, 0, 8, lane, f0);
  lb_quad_mfma(f1, bh, bl, &acc[4]);
  __builtin_amdgcn_sched_barrier(0);
  lb_quad_load<16>(buf, 0, 12, lane, f1);
  lb_quad_mfma(f0, bh, bl, &acc[8]);
  __builtin_amdgcn_sched_barrier(0);
  lb_quad_mfma(f1, bh, bl, &acc[12]);
  __builtin_amdgcn_sched_barrier(0);
}

// node's aggregated messages, 16-row layout (lane (n,g) owns chunks 4*mb + g of the 128-float row)
__device__ __forceinline__ void lb_load_agg16(const lb_node_args& a, int64_t gnode, int g, int k0, int k1,
                                              f32x4 (&v)[8]) {
  if (!a.fused) {
    const f32x4* gr = reinterpret_cast<const f32x4*>(a.agg) + gnode * 32 + g;
#pragma unroll
    for (int mb = 0; mb < 8; ++mb) v[mb] = gr[4 * mb];
    return;
  }
  const int t0 = k0 >> a.tile_shift, t1 = (k1 - 1) >> a.tile_shift;
  const bool single = t0 == t1;
  const int nsrc = (k1 <= k0) ? 0 : (single ? 1 : t1 - t0 + 1);
  auto slot_of = [&](int t) -> const f32x4* {
    const float* src = single ? a.agg + gnode * 128
                              : a.part + ((int64_t)t * 2 + (k0 <= (t << a.tile_shift) ? 0 : 1)) * 128;
    return reinterpret_cast<const f32x4*>(src) + g;
  };
  // the first two sources are fetched TOGETHER (a row of ~17 edges nearly always straddles one tile
  // boundary): summed source by source every source is its own dependent round trip - this kernel runs
  // the small launches, which are latency chains
  const f32x4* s0 = slot_of(t0);
  const f32x4* s1 = nsrc >= 2 ? slot_of(t0 + 1) : s0;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  f32x4 a1[8];
#pragma unroll
  for (int mb = 0; mb < 8; ++mb) {
    v[mb] = s0[4 * mb];
    a1[mb] = s1[4 * mb];
  }
#pragma unroll
  for (int mb = 0; mb < 8; ++mb) v[mb] = (nsrc >= 1 ? v[mb] : zero) + (nsrc >= 2 ? a1[mb] : zero);
  for (int s = 2; __any(s < nsrc); ++s) {
    if (s < nsrc) {
      const f32x4* s4 = slot_of(t0 + s);
#pragma unroll
      for (int mb = 0; mb < 8; ++mb) v[mb] = v[mb] + s4[4 * mb];
    }
  }
}

// NPA: k-steps (of 32) of input A (encoder features or node latents); NPB: 4 when the aggregated
// messages are a second input (processor), else 0.
// LOADERS (small launches, at most one workgroup per CU): four extra waves do nothing but move the weight
// chunks global -> registers -> LDS, two chunks in flight in their registers and a four-slot ring, so that a
// chunk has two full steps to arrive instead of one: a launch on a 2.5 k-particle graph is a latency chain of
// ten chunk steps (the four compute waves need ~0.4 us per chunk, an L2 round trip is > 1 us).
template <int NPA, int NPB, bool RESID, bool PROJ, bool LOADERS = false>
__global__ void __launch_bounds__(LOADERS ? 2 * N16_THREADS : N16_THREADS, LOADERS ? 1 : 2)
    k_node16h(lb_node_args a, const f32x4* __restrict__ w0h, const f32x4* __restrict__ w1h,
              const f32x4* __restrict__ wph) {
  constexpr int NSLOT = LOADERS ? 4 : 2;
  __shared__ f32x4 sB[NSLOT][CHUNK_VEC];
  __shared__ f32x4 sP[192];  // per-feature vectors, see below
  // the poison flag is READ first but only acted on once the first loads are in flight: branching on it here
  // would put a full round trip in front of every load of a launch that is a latency chain.  Only loads that
  // are in bounds whatever the state are issued before the check (rows, row_ptr, weights); the partial-sum slots
  // addressed THROUGH row_ptr are not (an overflowing step's row_ptr may point past the allocation).
  const int poisoned = a.ctrl->overflow_step;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = lane & 15, g = lane >> 4;
  constexpr int NP0 = NPA + NPB;
  constexpr int NCH0 = (NP0 + 1) / 2;  // chunks of W0 (2 k-steps x 8 blocks x hi|lo = 2048 vec)
  constexpr bool has_proj = PROJ;
  constexpr int n_chunks = NCH0 + 2 + (PROJ ? 4 : 0);

  // per-feature vectors into LDS: [0,32) b0, [32,64) b1, [64,96) ln_s, [96,128) ln_o, [128,192) bp
  if (tid < 128) {
    const float* src = tid < 32 ? a.b0 : (tid < 64 ? a.b1 : (tid < 96 ? a.ln_s : a.ln_o));
    sP[tid] = reinterpret_cast<const f32x4*>(src)[tid & 31];
  } else if (tid < 192) {
    sP[tid] = (has_proj && a.bp) ? reinterpret_cast<const f32x4*>(a.bp)[tid - 128] : f32x4{0.f, 0.f, 0.f, 0.f};
  }

  // chunk c -> (source, number of f32x4)
  auto chunk_src = [&](int c, const f32x4*& src, int& nvec) {
    if (c < NCH0) {
      src = w0h + (size_t)c * CHUNK_VEC;
      nvec = (2 * c + 2 <= NP0) ? CHUNK_VEC : CHUNK_VEC / 2;
    } else if (c < NCH0 + 2) {
      src = w1h + (size_t)(c - NCH0) * CHUNK_VEC;
      nvec = CHUNK_VEC;
    } else {
      src = wph + (size_t)(c - NCH0 - 2) * CHUNK_VEC;
      nvec = CHUNK_VEC;
    }
  };
  // Two-slot LDS ring, one staging register set: at step c the chunk c+1 (loaded during step c-1)
  // is committed into the slot chunk c-1 has just released, then the loads of chunk c+2 are issued
  // into the same registers, then chunk c is consumed; one barrier per step.  Two 4-wave workgroups
  // share a CU (2 x 67 KiB of LDS), so the prologue / epilogue memory phases of one overlap the
  // MFMA phase of the other.
  if constexpr (LOADERS) {
    if (wave >= N16_WAVES) {
      const int lt = tid - N16_THREADS;
      f32x4 set[2][N16_STG];
      auto issue = [&](int c) {
        const f32x4* src;
        int nvec;
        chunk_src(c, src, nvec);
#pragma unroll
        for (int i = 0; i < N16_STG; ++i) {
          const int idx = lt + i * N16_THREADS;
          set[c & 1][i] = src[idx < nvec ? idx : 0];
        }
      };
      auto commit = [&](int c) {
#pragma unroll
        for (int i = 0; i < N16_STG; ++i) sB[c & 3][lt + i * N16_THREADS] = set[c & 1][i];
      };
      issue(0);
      if (n_chunks > 1) issue(1);
      if (poisoned >= 0) return;
      commit(0);
      if (n_chunks > 1) commit(1);
      if (n_chunks > 2) issue(2);
      if (n_chunks > 3) issue(3);
      __syncthreads();
#pragma unroll
      for (int st = 0; st < n_chunks; ++st) {
        if (st + 2 < n_chunks) commit(st + 2);
        if (st + 4 < n_chunks) issue(st + 4);
        __syncthreads();
      }
      return;
    }
  }
  f32x4 stg[N16_STG];
  auto stage_issue = [&](int c) {
    if constexpr (LOADERS) return;
    const f32x4* src;
    int nvec;
    chunk_src(c, src, nvec);
#pragma unroll
    for (int i = 0; i < N16_STG; ++i) {
      const int idx = tid + i * N16_THREADS;
      stg[i] = src[idx < nvec ? idx : 0];
    }
  };
  auto stage_commit = [&](int c) {
    if constexpr (LOADERS) return;
#pragma unroll
    for (int i = 0; i < N16_STG; ++i) sB[c & 1][tid + i * N16_THREADS] = stg[i];
  };
  auto stage_step = [&](int c) {  // start of step c
    if (c + 1 < n_chunks) stage_commit(c + 1);
    if (c + 2 < n_chunks) stage_issue(c + 2);
  };

  // ---- this wave's 16 rows
  const int64_t row = ((int64_t)blockIdx.x * N16_WAVES + wave) * 16 + n;
  const bool valid = row < a.n_rows;
  const int64_t rowc = valid ? row : a.n_rows - 1;

  stage_issue(0);
  f32x4 va[2 * NPA];
  {
    const f32x4* xr = reinterpret_cast<const f32x4*>(a.xin) + rowc * (8 * NPA) + g;
#pragma unroll
    for (int mb = 0; mb < 2 * NPA; ++mb) va[mb] = xr[4 * mb];
  }
  f32x4 vb[NPB > 0 ? 8 : 1];
  int k0 = 0, k1 = 0;
  if (NPB > 0 && a.fused) {
    k0 = a.row_ptr[rowc];
    k1 = a.row_ptr[rowc + 1];
  }
  if (poisoned >= 0) return;
  if constexpr (NPB > 0) lb_load_agg16(a, rowc, g, k0, k1, vb);
  stage_commit(0);
  if (n_chunks > 1) stage_issue(1);
  __syncthreads();

  int c = 0;
  f32x4 acc[8];
#pragma unroll
  for (int mb = 0; mb < 8; ++mb) acc[mb] = sP[4 * mb + g];
  // ---- GEMM1 over [input A | aggregated messages]
#pragma unroll
  for (int ch = 0; ch < NCH0; ++ch, ++c) {
    stage_step(c);
    const f32x4* buf = sB[c & (NSLOT - 1)];
    h8 bh[2], bl[2];
#pragma unroll
    for (int pp = 0; pp < 2; ++pp) {
      const int p = 2 * ch + pp;
      if (p < NP0) {
        if (p < NPA)
          lb_split8n(va[2 * (p < NPA ? p : 0)], va[2 * (p < NPA ? p : 0) + 1], bh[pp], bl[pp]);
        else
          lb_split8n(vb[NPB > 0 ? 2 * (p - NPA) : 0], vb[NPB > 0 ? 2 * (p - NPA) + 1 : 0], bh[pp], bl[pp]);
      } else {
        bh[pp] = bh[0];
        bl[pp] = bl[0];
      }
    }
    lb_chunk_gemm8(buf, lane, (2 * ch + 1 < NP0) ? 2 : 1, bh[0], bl[0], bh[1], bl[1], acc);
    __syncthreads();
  }
#pragma unroll
  for (int mb = 0; mb < 8; ++mb)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[mb][j] = fmaxf(acc[mb][j], 0.f);
  // ---- GEMM2
  f32x4 acc2[8];
#pragma unroll
  for (int mb = 0; mb < 8; ++mb) acc2[mb] = sP[32 + 4 * mb + g];
#pragma unroll
  for (int ch = 0; ch < 2; ++ch, ++c) {
    stage_step(c);
    const f32x4* buf = sB[c & (NSLOT - 1)];
    h8 bh[2], bl[2];
#pragma unroll
    for (int pp = 0; pp < 2; ++pp) lb_split8n(acc[2 * (2 * ch + pp)], acc[2 * (2 * ch + pp) + 1], bh[pp], bl[pp]);
    lb_chunk_gemm8(buf, lane, 2, bh[0], bl[0], bh[1], bl[1], acc2);
    __syncthreads();
  }
  // ---- LayerNorm (+ residual)
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
    const f32x4 sc = sP[64 + 4 * mb + g], of = sP[96 + 4 * mb + g];
#pragma unroll
    for (int j = 0; j < 4; ++j) y[mb][j] = (sc[j] * rs) * (acc2[mb][j] - mean) + of[j];
    if constexpr (RESID) y[mb] = va[mb < 2 * NPA ? mb : 0] + y[mb];
  }
  if (valid) {
    f32x4* nr = reinterpret_cast<f32x4*>(a.nlat) + row * 32 + g;
#pragma unroll
    for (int mb = 0; mb < 8; ++mb) nr[4 * mb] = y[mb];
  }
  // ---- projection for the next edge MLP: psr = y @ [Ws | Wr] + [0 | b0_next], 16 output blocks
  if (has_proj) {
    f32x4 accp[16];
#pragma unroll
    for (int mb = 0; mb < 16; ++mb) accp[mb] = sP[128 + 4 * mb + g];
#pragma unroll
    for (int p = 0; p < 4; ++p, ++c) {
      stage_step(c);
      const f32x4* buf = sB[c & (NSLOT - 1)];
      h8 bh, bl;
      lb_split8n(y[2 * p], y[2 * p + 1], bh, bl);
      lb_chunk_proj16(buf, lane, bh, bl, accp);
      __syncthreads();
    }
    if (valid) {
      f32x4* pr = reinterpret_cast<f32x4*>(a.psr) + row * 64 + g;
#pragma unroll
      for (int mb = 0; mb < 16; ++mb) pr[4 * mb] = accp[mb];
    }
  }
}

int lbk_node16h(lb_engine* e, const lb_node_args& a, const float* w0h, const float* w1h,
                const float* wph, int npa, int npb, bool resid) {
  const int nblk = (int)((a.n_rows + 16 * N16_WAVES - 1) / (16 * N16_WAVES));
  const f32x4* w0 = reinterpret_cast<const f32x4*>(w0h);
  const f32x4* w1 = reinterpret_cast<const f32x4*>(w1h);
  const f32x4* wp = reinterpret_cast<const f32x4*>(wph);
  // LB_NODE_LOADERS=0: no loader waves
  static const bool want_loaders = !(getenv("LB_NODE_LOADERS") && getenv("LB_NODE_LOADERS")[0] == '0');
  const bool loaders = want_loaders && nblk <= 256;
  dim3 grid(nblk), block(loaders ? 2 * N16_THREADS : N16_THREADS);
  const bool proj = wph != nullptr;
#define LB_N16(A, B, R)                                                                                 \
  do {                                                                                                  \
    if (proj && loaders)                                                                                \
      hipLaunchKernelGGL((k_node16h<A, B, R, true, true>), grid, block, 0, e->stream, a, w0, w1, wp);   \
    else if (proj)                                                                                      \
      hipLaunchKernelGGL((k_node16h<A, B, R, true>), grid, block, 0, e->stream, a, w0, w1, wp);         \
    else if (loaders)                                                                                   \
      hipLaunchKernelGGL((k_node16h<A, B, R, false, true>), grid, block, 0, e->stream, a, w0, w1, wp);  \
    else                                                                                                \
      hipLaunchKernelGGL((k_node16h<A, B, R, false>), grid, block, 0, e->stream, a, w0, w1, wp);        \
  } while (0)
  if (npa == 4 && npb == 4 && resid)
    LB_N16(4, 4, true);
  else if (npa == 1 && npb == 0 && !resid)
    LB_N16(1, 0, false);
  else if (npa == 2 && npb == 0 && !resid)
    LB_N16(2, 0, false);
  else if (npa == 3 && npb == 0 && !resid)
    LB_N16(3, 0, false);
  else if (npa == 4 && npb == 0 && !resid)
    LB_N16(4, 0, false);
  else
    return lb_fail(LB_ERR_UNSUPPORTED, "k_node16h<%d,%d,%d> not instantiated", npa, npb, (int)resid);
  LB_HIP(hipGetLastError());
  return LB_OK;
}
