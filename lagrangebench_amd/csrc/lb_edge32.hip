// lb_edge32.hip - processor edge kernel on 32-edge tiles (round 3).
//
// Same mathematics and HBM layouts as k_edge16v (lb_edge16v.hip; reference: GNS._processor update_edge_features +
// jraph.segment_sum, models/gns.py:86-122): f16x2 arithmetic (fp32 operand = fp16 hi + fp16 lo, products
// lo*hi + hi*lo + hi*hi accumulated in fp32), LayerNorm + residual, fused segmented sum over the receivers.  What
// changes is the tile: one wave owns TWO consecutive 16-edge tiles (2T, 2T+1 of the tile-blocked latent layout) and
// multiplies on v_mfma_f32_32x32x16_f16 (A = 32 output features x 16 k, B = 16 k x 32 edges):
//   * every weight fragment read from LDS now serves 32 edges: 128 ds_read_b128 per 32 edges instead of 2 x 152;
//   * the per-tile bookkeeping (index / segment arithmetic, probes, LDS vector reads, address formation) is paid once
//     per 32 edges; the LayerNorm reduction is one lane-pair exchange instead of two;
//   * a wave keeps 16 KiB of latents + 32 KiB of gathers in flight.
// Lane (n32 = lane & 31, h = lane >> 5) holds edge 32T + n32; its 16-edge tile is t16 = 2T + (n32 >> 4), its DPP row
// (16 lanes) is exactly that tile's 16 edges for one h - the segmented scan, the agg / part slot rules and therefore
// the summation ORDER are those of k_edge16v: the aggregated messages are bit-identical to its.
// Register layout: 16 "slots" of one f32x4 per lane; slot qs = 2 mb + e holds features 16 mb + 4 (h + 2 e) + {0..3}
// (mb = 0..7, e = 0, 1) = quad (mb, g = h + 2e) of the 16-edge tile layout.  Slots 2 kb, 2 kb + 1 are the 8 B-operand
// values of k-step kb (k = 8 h + j  <->  feature 16 kb + 8 (j >> 2) + 4 h + (j & 3): a permutation of the k order that
// is folded into the packed weights), and slots 4 q .. 4 q + 3 are the C registers of output block q (rows
// 8 (r >> 2) + 4 h + (r & 3)): a GEMM's output is the next GEMM's operand without any data movement.
//
// MEASURED (MI355X, TGV3D-8k x 8, profiles/r03_edge32.txt): parity green (tests/test_switches_gpu.py, LB_EDGE32=1), but
// SLOWER than k_edge16v - 300 vs 257 us per launch (LDC3D-8k B = 1: 0.72 vs 0.66 ms/step).  Ablation in place (ABL
// template parameter): without the gathers 262 us, without the GEMMs 239, without both 207; k_edge16v loses only ~25 us
// to its GEMMs.  The instruction count per edge is the same (1282 vs 1332 VALU per 32 edges: scan, split, ReLU and
// LayerNorm are per element), the LDS weight reads that did halve were not the limiter, and a wave now alternates
// between a 48-load burst and ~6 k cycles of back-to-back MFMAs, which the second wave of the SIMD covers less well
// than two 16-edge tiles did.  Opt-in (LB_EDGE32=1); the default stays k_edge16v.
#include <stdlib.h>
#include <string.h>

#include "lb_f16x2.h"

#define MFMA32H(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)

static inline uint16_t e32_f32_to_f16(float x) {
  _Float16 h = (_Float16)x;  // RNE
  uint16_t u;
  memcpy(&u, &h, 2);
  return u;
}
static inline float e32_f16_to_f32(uint16_t u) {
  _Float16 h;
  memcpy(&h, &u, 2);
  return (float)h;
}

// out: K*M floats worth of storage holding [(kb*(M/32) + q)][part: 0 hi, 1 lo][lane][8 halfs]; w row-major [K][M]
void lb_pack_weight32h(const float* w, int K, int M, float* out) {
  uint16_t* o = reinterpret_cast<uint16_t*>(out);
  const int NKB = K / 16, NQ = M / 32;
  for (int kb = 0; kb < NKB; ++kb)
    for (int q = 0; q < NQ; ++q)
      for (int lane = 0; lane < 64; ++lane)
        for (int j = 0; j < 8; ++j) {
          const int h = lane >> 5, m = 32 * q + (lane & 31);
          const int k = 16 * kb + 8 * (j >> 2) + 4 * h + (j & 3);
          const float x = w[(size_t)k * M + m];
          const uint16_t hi = e32_f32_to_f16(x);
          const uint16_t lo = e32_f32_to_f16(x - e32_f16_to_f32(hi));
          const size_t base = ((size_t)(kb * NQ + q) * 2) * 64;
          o[(base + lane) * 8 + j] = hi;
          o[(base + 64 + lane) * 8 + j] = lo;
        }
}

// acc[q] (+)= W^T * v over 8 k-steps of 16, f16x2, phase-pipelined like lb_gemm16v: the `lo` fragments of k-step
// kb+1 are fetched while the eight `hi` MFMAs of kb run and the `hi` fragments while its four `lo` MFMAs run; an
// accumulator is touched again only after three other MFMAs.  wbase: this lane's LDS pointer to fragment (0, 0, hi).
// vq(qs): slot qs (f32x4) of the B operand (an array, or the previous GEMM's accumulators in place).
// hook(): called once after k-step 4 has consumed its operand slots (the caller's late loads go out there).
template <bool RELU, bool GUARD, typename VQ, typename HOOK>
__device__ __forceinline__ void lb_gemm32(lds_cptr wbase, VQ vq, f32x16 (&acc)[4], uint32_t* orv, HOOK hook) {
  auto note = [&](const h8& hh) {
    if constexpr (GUARD) {
      typedef uint32_t u32x4g __attribute__((ext_vector_type(4)));
      const u32x4g u = __builtin_bit_cast(u32x4g, hh);
      *orv |= (u[0] | u[1]) | (u[2] | u[3]);
    }
  };
  auto frag = [&](int kb, int q, int part) -> h8 { return __builtin_bit_cast(h8, wbase[((kb * 4 + q) * 2 + part) * 64]); };
  auto relu4 = [&](const f32x4& x) -> f32x4 {
    if (!RELU) return x;
    f32x4 r;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float f = x[j];
      r[j] = __builtin_bit_cast(float, max(__builtin_bit_cast(int, f), 0));
    }
    return r;
  };
  h8 X[4], Y[4], bh, bl, nbh, nbl;
#pragma unroll
  for (int q = 0; q < 4; ++q) X[q] = frag(0, q, 1);
#pragma unroll
  for (int q = 0; q < 4; ++q) Y[q] = frag(0, q, 0);
  lb_split8v(relu4(vq(0)), relu4(vq(1)), bh, bl);
  note(bh);
  SB();
#pragma unroll
  for (int kb = 0; kb < 8; ++kb) {
    __builtin_amdgcn_s_waitcnt(LB_WAIT_LGKM(4));
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = MFMA32H(X[q], bh, acc[q]);
    SB();
    if (kb < 7) {
#pragma unroll
      for (int q = 0; q < 4; ++q) X[q] = frag(kb + 1, q, 1);
      lb_split8v(relu4(vq(2 * kb + 2)), relu4(vq(2 * kb + 3)), nbh, nbl);
      note(nbh);
      if (kb == 4) hook();
      __builtin_amdgcn_s_waitcnt(LB_WAIT_LGKM(4));
    } else {
      __builtin_amdgcn_s_waitcnt(LB_WAIT_LGKM(0));
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = MFMA32H(Y[q], bl, acc[q]);
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = MFMA32H(Y[q], bh, acc[q]);
    SB();
    if (kb < 7) {
#pragma unroll
      for (int q = 0; q < 4; ++q) Y[q] = frag(kb + 1, q, 0);
      bh = nbh;
      bl = nbl;
    }
    SB();
  }
}

// slot qs of a block-of-16-registers accumulator set
__device__ __forceinline__ f32x4 e32_quad(const f32x16 (&a)[4], int qs) {
  const f32x16& b = a[qs >> 2];
  const int r = 4 * (qs & 3);
  return f32x4{b[r], b[r + 1], b[r + 2], b[r + 3]};
}
__device__ __forceinline__ void e32_set_quad(f32x16 (&a)[4], int qs, const f32x4& v) {
  f32x16& b = a[qs >> 2];
  const int r = 4 * (qs & 3);
  b[r] = v[0];
  b[r + 1] = v[1];
  b[r + 2] = v[2];
  b[r + 3] = v[3];
}

// sum over the lane pair (n32, 0), (n32, 1): gfx950 v_permlane32_swap (inline asm, see lb_msplit_dev.h)
__device__ __forceinline__ float e32_pair_sum(float x) {
  float a = x, b = x;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  return a + b;
}

// NT: the edge latents are streamed with nontemporal loads / stores (batches whose latents exceed the Infinity Cache)
// GUARD: exhaustive TINY test of the f16x2 range guard (lb_tile_tiny), else the sampled probe.  SKIP: last layer.
// RELOAD: the last RELOAD of the 16 latent slots are not kept for the residual but read a second time after the GEMMs
// (a few us after the first read: L2 / Infinity Cache) - the second GEMM's live set (latents 64 + hidden 64 +
// accumulators 64 + fragments 48 registers) otherwise spills a handful of registers to scratch at 256 VGPRs.
template <bool SKIP, bool NT, bool GUARD, int RELOAD = 2, int ABL = 0>
__global__ void __launch_bounds__(512, 2) k_edge32(lb_edge16_args a) {
  constexpr int THREADS = 512, WAVES = 8, NW0 = 4096;
  __shared__ f32x4 sW[NW0 + 4096 + 96];
  const int poisoned = a.ctrl->overflow_step;
  const int E = a.ctrl->n_edges_total;
  const float ln_inv_d = a.ctrl->ln_inv_d, ln_pad = a.ctrl->ln_pad;
  const int tid = threadIdx.x;
  constexpr int NST = (NW0 + 4096 + THREADS - 1) / THREADS;
  f32x4 st[NST];
  {
    const f32x4* g0 = reinterpret_cast<const f32x4*>(a.w0p);
    const f32x4* g1 = reinterpret_cast<const f32x4*>(a.w1p);
#pragma unroll
    for (int k = 0; k < NST; ++k) {
      const int i = tid + k * THREADS;
      st[k] = i < NW0 ? g0[i] : g1[(i < NW0 + 4096 ? i : NW0 + 4095) - NW0];
    }
  }
  const int ntiles16 = (E + 15) >> 4, ntiles = (ntiles16 + 1) >> 1;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n32 = lane & 31, h = lane >> 5, sub = (lane >> 4) & 1, n = lane & 15;
  const int xcd = blockIdx.x & 7, slot = (blockIdx.x >> 3) * WAVES + wave;
  const int stride = (gridDim.x >> 3) * WAVES;
  const int t_lo = (int)(((int64_t)ntiles * xcd) >> 3), t_hi = (int)(((int64_t)ntiles * (xcd + 1)) >> 3);
  int t = t_lo + slot;
  auto rowc_of = [&](int tt) -> int64_t {
    const int row = tt * 32 + n32;
    return row < E ? row : (E > 0 ? E - 1 : 0);
  };
  int s_c = 0, r_c = 0;
  if (t < t_hi) {
    const int64_t rc = rowc_of(t);
    s_c = a.senders[rc];
    r_c = a.receivers[rc];
  }
  if (poisoned >= 0) return;
  {
#pragma unroll
    for (int k = 0; k < NST; ++k) {
      const int i = tid + k * THREADS;
      if (i < NW0 + 4096) sW[i] = st[k];
    }
    if (tid < 96) {
      const float* src = tid < 32 ? a.b1 : (tid < 64 ? a.ln_s : a.ln_o);
      sW[NW0 + 4096 + tid] = reinterpret_cast<const f32x4*>(src)[tid & 31];
    }
  }
  __syncthreads();
  if (t >= t_hi) return;
  uint32_t off0 = (uint32_t)(uintptr_t)(lds_cptr)(sW + lane);
  uint32_t off1 = (uint32_t)(uintptr_t)(lds_cptr)(sW + NW0 + lane);
  uint32_t off2 = (uint32_t)(uintptr_t)(lds_cptr)(sW + NW0 + 4096 + h);
  asm volatile("" : "+v"(off0), "+v"(off1), "+v"(off2));
  const lds_cptr w0b = (lds_cptr)(uintptr_t)off0, w1b = (lds_cptr)(uintptr_t)off1, vecb = (lds_cptr)(uintptr_t)off2;
  const f32x4* psr4 = reinterpret_cast<const f32x4*>(a.psr);
  const int n_iter = (t_hi - 1 - t) / stride + 1;
  const int t_last = t + (n_iter - 1) * stride;
  asm volatile("" : "+v"(s_c), "+v"(r_c));
  int guard_tiny = 0;
  for (int it = 0; it < n_iter; ++it, t += stride) {
    const int t16 = 2 * t + sub;  // this lane's 16-edge tile (one past the last one for an odd tile count: inside
                                  // the padded allocation, every edge of it invalid)
    f32x4 ve[16];
    f32x16 acc[4];
    const int r_cur = r_c;
    // quad (mb, g = h + 2e) of tile t16: elat4[(t16 * 8 + mb) * 64 + g * 16 + n]
    const f32x4* er = reinterpret_cast<const f32x4*>(a.elat) + (int64_t)t16 * 512 + h * 16 + n;
    {
      const f32x4* ps = psr4 + (int64_t)s_c * 64 + h;
      const f32x4* pr = psr4 + (int64_t)r_c * 64 + 32 + h;
      f32x4 p0[16], p1[16];
#pragma unroll
      for (int qs = 0; qs < 16; ++qs) {
        const int mb = qs >> 1, e = qs & 1;
        ve[qs] = NT ? __builtin_nontemporal_load(&er[64 * mb + 32 * e]) : er[64 * mb + 32 * e];
        p0[qs] = (ABL & 1) ? f32x4{.1f, .2f, .05f * (float)(s_c & 7), .01f * (float)mb} : ps[4 * mb + 2 * e];
        p1[qs] = (ABL & 1) ? f32x4{.3f, .1f, .05f * (float)(r_c & 7), .01f * (float)mb} : pr[4 * mb + 2 * e];
      }
      const int64_t rn = rowc_of(min(t + stride, t_last));
      s_c = a.senders[rn];
      r_c = a.receivers[rn];
#pragma unroll
      for (int qs = 0; qs < 16; ++qs) e32_set_quad(acc, qs, lb_pk_add(p1[qs], p0[qs]));
    }
    int rb = lb_edge_probe(a.receivers, t16, lane, E);
    if (it == 0) lb_range_probe(a.ctrl, ve, 16);
    __builtin_amdgcn_s_setprio(2);
    uint32_t or_e = 0, or_h = 0;
    if constexpr (!(ABL & 8)) lb_gemm32<false, GUARD>(w0b, [&](int qs) -> f32x4 { return ve[qs]; }, acc, &or_e, [] {});
    if (it == 0) {
      f32x4 hid[16];
#pragma unroll
      for (int qs = 0; qs < 16; ++qs) hid[qs] = e32_quad(acc, qs);
      lb_range_probe(a.ctrl, hid, 16);
    }
    f32x16 acc2[4];
#pragma unroll
    for (int qs = 0; qs < 16; ++qs) {
      const f32x4 bv = vecb[4 * (qs >> 1) + 2 * (qs & 1)];
      e32_set_quad(acc2, qs, bv);
    }
    if constexpr (!(ABL & 8)) {
      // the re-read latent slots go out in the middle of the second GEMM (into registers the consumed hidden slots
      // have freed): issued after it they put an L2 round trip in front of the stores
      lb_gemm32<true, GUARD>(w1b, [&](int qs) -> f32x4 { return e32_quad(acc, qs); }, acc2, &or_h, [&] {
        if constexpr (!SKIP && RELOAD > 0) {
#pragma unroll
          for (int qs = 16 - RELOAD; qs < 16; ++qs) ve[qs] = er[64 * (qs >> 1) + 32 * (qs & 1)];
        }
      });
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) acc2[q] = acc2[q] + acc[q];
    }
    __builtin_amdgcn_s_setprio(0);
    if constexpr (GUARD) guard_tiny |= (int)lb_tile_tiny(or_e) | (int)lb_tile_tiny(or_h);
    // take delivery of the next tile's indices while only loads are in flight (in-order vmcnt)
    asm volatile("" : "+v"(s_c), "+v"(r_c), "+v"(rb));
    // ---- LayerNorm over the 128 features of every edge: 64 values here, 64 in the partner lane (lane ^ 32)
    f32x4 y[16];
    {
      f32x2v s2 = {0.f, 0.f};
#pragma unroll
      for (int qs = 0; qs < 16; ++qs) {
        y[qs] = e32_quad(acc2, qs);
        s2 = s2 + (lb_lo2(y[qs]) + lb_hi2(y[qs]));
      }
      const float mean = e32_pair_sum(s2[0] + s2[1]) * ln_inv_d;
      const f32x2v m2 = {mean, mean};
      f32x2v v2 = {0.f, 0.f};
#pragma unroll
      for (int qs = 0; qs < 16; ++qs) {
        const f32x2v dl = lb_lo2(y[qs]) - m2, dh = lb_hi2(y[qs]) - m2;
        v2 = __builtin_elementwise_fma(dl, dl, v2);
        v2 = __builtin_elementwise_fma(dh, dh, v2);
        y[qs] = lb_cat2(dl, dh);
      }
      const float vs = e32_pair_sum(v2[0] + v2[1]);
      const float rs = 1.0f / sqrtf(fmaxf(vs - ln_pad * (mean * mean), 0.f) * ln_inv_d + 1e-5f);
      const f32x2v r2 = {rs, rs};
#pragma unroll
      for (int qs = 0; qs < 16; ++qs) {
        const f32x4 sc = vecb[32 + 4 * (qs >> 1) + 2 * (qs & 1)], of = vecb[64 + 4 * (qs >> 1) + 2 * (qs & 1)];
        y[qs] = lb_cat2(__builtin_elementwise_fma(lb_lo2(sc) * r2, lb_lo2(y[qs]), lb_lo2(of)),
                        __builtin_elementwise_fma(lb_hi2(sc) * r2, lb_hi2(y[qs]), lb_hi2(of)));
      }
    }
    const int row = t * 32 + n32;
    const bool valid = row < E;
    if constexpr (!SKIP) {
      f32x4* ew = reinterpret_cast<f32x4*>(a.elat_out ? a.elat_out : a.elat) + (int64_t)t16 * 512 + h * 16 + n;
#pragma unroll
      for (int qs = 0; qs < 16; ++qs) {
        const int mb = qs >> 1, e = qs & 1;
        if constexpr (NT)
          __builtin_nontemporal_store(lb_pk_add(ve[qs], y[qs]), &ew[64 * mb + 32 * e]);
        else
          ew[64 * mb + 32 * e] = lb_pk_add(ve[qs], y[qs]);
      }
    }
    // ---- fused jraph.segment_sum: segmented Hillis-Steele scan inside each 16-lane DPP row (= one 16-edge tile)
    const int rr = valid ? r_cur : (-1 - n);
    const int r_prev = __builtin_amdgcn_update_dpp(-2, rr, 0x111, 0xF, 0xF, false);
    const bool head = (n == 0) || (rr != r_prev);
    const unsigned H = (unsigned)((__ballot(head) >> (lane & 48)) & 0xffffull);
    const unsigned below = H & ((2u << n) - 1u);
    const int segstart = 31 - __clz(below);
    const bool tail = (n == 15) || ((H >> (n + 1)) & 1u);
    const float m1 = (n >= 1 && segstart <= n - 1) ? 1.f : 0.f, m2s = (n >= 2 && segstart <= n - 2) ? 1.f : 0.f;
    const float m4 = (n >= 4 && segstart <= n - 4) ? 1.f : 0.f, m8 = (n >= 8 && segstart <= n - 8) ? 1.f : 0.f;
    if (!valid) {
#pragma unroll
      for (int qs = 0; qs < 16; ++qs) y[qs] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int qs = 0; qs < 16; qs += 2) lb_scan8(y[qs], y[qs + 1], m1, m2s, m4, m8);
    // lb_seg_complete with the row's own probe lanes (lane 0 / 1 of the DPP row hold the receivers of the edge before /
    // after this 16-edge tile); exchanged while every lane is active
    const int r_before = __shfl(rb, lane & 48), r_after = __shfl(rb, (lane & 48) | 1);
    if (tail && valid) {
      const bool starts_before = segstart == 0 && t16 > 0 && r_before == rr;
      const bool ends_after = n == 15 && t16 * 16 + 16 < E && r_after == rr;
      const int slot01 = segstart == 0 ? 0 : 1;
      const bool complete = !starts_before && !ends_after;
      float* dst = complete ? a.agg + (int64_t)rr * 128 : a.part + ((int64_t)t16 * 2 + slot01) * 128;
      f32x4* d4 = reinterpret_cast<f32x4*>(dst) + h;
#pragma unroll
      for (int qs = 0; qs < 16; ++qs) d4[4 * (qs >> 1) + 2 * (qs & 1)] = y[qs];
    }
  }
  if (guard_tiny && lane == 0) lb_raise_math(a.ctrl, LB_MATH_TINY);
}

int lbk_edge32(lb_engine* e, const lb_edge16_args& a) {
#define LB_E32_(NT, G, GU)                                                         \
  do {                                                                             \
    if (a.skip_elat_store)                                                         \
      LB_LAUNCH_TIMED(e, (k_edge32<true, NT, GU>), dim3(G), dim3(512), a);         \
    else                                                                           \
      LB_LAUNCH_TIMED(e, (k_edge32<false, NT, GU>), dim3(G), dim3(512), a);        \
  } while (0)
#define LB_E32(NT, G)        \
  do {                       \
    if (e->guard_full)       \
      LB_E32_(NT, G, true);  \
    else                     \
      LB_E32_(NT, G, false); \
  } while (0)
  const int64_t tiles16 = ((int64_t)e->e_cap * e->g.B + 15) / 16, tiles_cap = (tiles16 + 1) / 2;
  int64_t g = (tiles_cap + 7) / 8;
  g = (g + 7) / 8 * 8;
  static const int grid_cap = getenv("LB_EDGE_GRID") ? atoi(getenv("LB_EDGE_GRID")) : 256;
  const int grid = (int)(g < 8 ? 8 : (g > grid_cap ? grid_cap : g));
  static const int64_t nt_min_tiles = getenv("LB_EDGE_NT_MIN_TILES") ? atoll(getenv("LB_EDGE_NT_MIN_TILES")) : 12288;
  if (tiles16 < nt_min_tiles)
    LB_E32(false, grid);
  else
    LB_E32(true, grid);
#undef LB_E32
#undef LB_E32_
  LB_HIP(hipGetLastError());
  return LB_OK;
}
