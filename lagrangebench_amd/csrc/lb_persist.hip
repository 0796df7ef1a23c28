// lb_persist.hip - the GNS processor (all L message-passing layers) as ONE persistent launch for small graphs.
//
// Reference: GNS._processor, lagrangebench/models/gns.py:83-124 - L x [update_edge_features -> segment_sum ->
// update_node_features -> residuals].  Every layer is two grid-wide dependencies (node latents of ALL senders before
// the edge update, ALL messages of a receiver before the node update).  As 2 L launches (lb_msplit.hip) every one of
// them pays a kernel boundary plus its own start-up chain "control block -> indices -> gathers -> weights": on one
// 2.5 k-particle trajectory ~10 us per launch of which ~2 us is arithmetic (profiles/r03_tgv2d_b1_kernel_trace.txt).
// Here one workgroup per CU stays resident for the whole processor:
//   * phases (edge k, node k) are separated by a counter barrier over the grid (per-XCD-group arrival counters ->
//     top counter -> generation word, relaxed agent-scope atomics, bounded spin);
//   * what crosses workgroups (psr = node projections, agg / part = aggregated messages) is written WRITE-THROUGH
//     (agent-scope 8-byte stores = global_store_dwordx2 sc1; every wave drains its stores before the arrival), ONE
//     lane per workgroup issues an agent-scope acquire after the barrier, then everybody reads with plain 16-byte
//     loads (cdna_hip_programming.md Guideline 16, recipe R1).  Measured on the way (round 3): sc1 stores + sc1 LOADS
//     without the acquire return stale rows (deterministically wrong results) although each side works when paired
//     with the plain + fence form; 16-byte raw-buffer sc1 accesses gave the same wrong results;
//   * tile -> workgroup assignment is static, so edge latents and node latents are only ever touched by the
//     workgroup that wrote them (plain accesses);
//   * the weights of the NEXT phase are loaded into registers while the workgroup waits at the barrier (edge MLP: 128
//     VGPRs per wave, node MLP + projection: 320; the two sets share registers - their live ranges do not overlap);
//     the indices / CSR bounds / own latents of a phase's first tile are requested BEFORE the barrier too.
// Tile bodies = the M-split kernels' (lb_msplit_dev.h).  Safety: every spin is bounded; a time-out raises
// lb_ctrl::persist_error, every workgroup leaves, the host reports LB_ERR_STATE and the engine goes back to the
// multi-launch path.
#include <stdio.h>
#include <stdlib.h>

#include "lb_msplit_dev.h"


// Cross-workgroup payload: write-through stores (two 8-byte agent-scope relaxed stores per 16 bytes), plain loads
// behind the barrier's acquire.
struct ps_buf {
  unsigned long long* p;
};
__device__ __forceinline__ ps_buf ps_mk(const void* p) { return ps_buf{(unsigned long long*)const_cast<void*>(p)}; }
__device__ __forceinline__ f32x4 ps_ld(const ps_buf& b, int64_t f4_index) {
  return reinterpret_cast<const f32x4*>(b.p)[f4_index];
}
__device__ __forceinline__ void ps_st(const ps_buf& b, int64_t f4_index, const f32x4& v) {
  union {
    unsigned long long u[2];
    f32x4 v;
  } x;
  x.v = v;
  __hip_atomic_store(b.p + 2 * f4_index, x.u[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(b.p + 2 * f4_index + 1, x.u[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

#define PS_BAR_STRIDE 32  // words: every barrier word on its own 128-byte line
#define PS_SPIN_LIMIT (1u << 21)

// Grid barrier.  bar: [0..7] group arrival counters, [8] top counter, [9] generation, [10] error (each * stride);
// monotonic within a launch (zeroed by the host before it), epoch = 1, 2, ...
__device__ __forceinline__ bool ps_grid_barrier(unsigned* bar, unsigned epoch, unsigned n_in_group, int* s_ok,
                                                lb_ctrl* ctrl) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // EVERY wave drains its write-through stores
  __syncthreads();
  if (threadIdx.x == 0) {
    int ok = 1;
    const unsigned grp = blockIdx.x & 7;
    const unsigned old = __hip_atomic_fetch_add(&bar[grp * PS_BAR_STRIDE], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old + 1 == epoch * n_in_group) {
      const unsigned o2 = __hip_atomic_fetch_add(&bar[8 * PS_BAR_STRIDE], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (o2 + 1 == epoch * 8) __hip_atomic_store(&bar[9 * PS_BAR_STRIDE], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    unsigned spins = 0;
    while (__hip_atomic_load(&bar[9 * PS_BAR_STRIDE], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > PS_SPIN_LIMIT ||
          __hip_atomic_load(&bar[10 * PS_BAR_STRIDE], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
        __hip_atomic_store(&bar[10 * PS_BAR_STRIDE], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        atomicExch(&ctrl->persist_error, 1);
        atomicMin(&ctrl->persist_step, ctrl->step);
        ok = 0;
        break;
      }
    }
    *s_ok = ok;
    // ONE agent-scope acquire per workgroup drops the stale L1 / L2 lines; the payload is then read with plain loads
    asm volatile("; acquire" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
  asm volatile("" ::: "memory");
  return *s_ok != 0;
}

__global__ void __launch_bounds__(MS_THREADS, 1) k_gns_persist(lb_persist_args a) {
  __shared__ f32x4 sB1[8 * 2 * 64];  // edge: 4 k-blocks, node: 8
  __shared__ f32x4 sB2[4 * 2 * 64];
  __shared__ f32x4 sB3[4 * 2 * 64];
  __shared__ __attribute__((aligned(16))) f32x2m sRed[16 * 4];
  __shared__ __attribute__((aligned(16))) float sMx[3][4];
  __shared__ int sOk;
  const int poisoned = a.ctrl->overflow_step;
  const int E = a.ctrl->n_edges_total;
  const float ln_inv_d = a.ctrl->ln_inv_d, ln_pad = a.ctrl->ln_pad;
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, g = lane >> 4;
  if (poisoned >= 0) return;  // uniform over the GRID (nobody writes the control block during this launch)
  const int ntiles_e = (E + 15) >> 4;
  const int ntiles_n = (int)((a.n_rows + 15) >> 4);
  ms_walk we, wn;
  const bool has_e = we.init(ntiles_e), has_n = wn.init(ntiles_n);
  const unsigned n_in_group = gridDim.x >> 3;
  const ps_buf r_psr = ps_mk(a.psr), r_agg = ps_mk(a.agg), r_part = ps_mk(a.part);
  const f32x4* elat4 = reinterpret_cast<const f32x4*>(a.elat);
  const f32x4* nlat4 = reinterpret_cast<const f32x4*>(a.nlat);
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  ps_guard guard{0.f, 0};
  unsigned epoch = 0;

  // edge-phase prefetch state
  f32x4 nve[2], nps[2], npr[2];
  int nr = 0, nrb = 0, s_i = 0, r_i = 0;
  auto load_idx = [&](int t) {
    const int row = t * 16 + n;
    const int rc = row < E ? row : E - 1;
    s_i = a.senders[rc];
    r_i = a.receivers[rc];
  };
  auto issue_own = [&](int t) {  // this workgroup's own latents: plain
    nve[0] = elat4[((int64_t)t * 8 + 2 * w) * 64 + lane];
    nve[1] = elat4[((int64_t)t * 8 + 2 * w + 1) * 64 + lane];
    nr = r_i;
    nrb = lb_edge_probe(a.receivers, t, lane, E);
  };
  auto issue_psr = [&]() {  // other workgroups' projections: sc1
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      nps[c] = ps_ld(r_psr, (int64_t)s_i * 64 + 8 * w + 4 * c + g);
      npr[c] = ps_ld(r_psr, (int64_t)r_i * 64 + 32 + 8 * w + 4 * c + g);
    }
  };
  // node-phase inputs of one tile
  f32x4 xa[2], ag[2];
  int nk0 = 0, nk1 = 0;
  auto node_rows_own = [&](int t, int64_t& rc, bool& valid) {  // own latents + CSR bounds: before the barrier
    const int64_t row = (int64_t)t * 16 + n;
    valid = row < a.n_rows;
    rc = valid ? row : a.n_rows - 1;
    xa[0] = nlat4[rc * 32 + 8 * w + g];
    xa[1] = nlat4[rc * 32 + 8 * w + 4 + g];
    nk0 = a.row_ptr[rc];
    nk1 = a.row_ptr[rc + 1];
  };
  auto node_agg = [&](int64_t rc) {  // other workgroups' messages: sc1, after the barrier
    const int t0 = nk0 >> 4, t1 = (nk1 - 1) >> 4;
    const bool single = t0 == t1;
    const int nsrc = (nk1 <= nk0) ? 0 : (single ? 1 : t1 - t0 + 1);
    auto ld2 = [&](int tt, f32x4& v0, f32x4& v1) {
      if (single) {
        v0 = ps_ld(r_agg, rc * 32 + 8 * w + g);
        v1 = ps_ld(r_agg, rc * 32 + 8 * w + 4 + g);
      } else {
        const int64_t base = ((int64_t)tt * 2 + (nk0 <= (tt << 4) ? 0 : 1)) * 32 + 8 * w + g;
        v0 = ps_ld(r_part, base);
        v1 = ps_ld(r_part, base + 4);
      }
    };
    f32x4 v00, v01, v10, v11;
    ld2(t0, v00, v01);
    ld2(nsrc >= 2 ? t0 + 1 : t0, v10, v11);
    ag[0] = (nsrc >= 1 ? v00 : zero) + (nsrc >= 2 ? v10 : zero);
    ag[1] = (nsrc >= 1 ? v01 : zero) + (nsrc >= 2 ? v11 : zero);
    for (int s = 2; __any(s < nsrc); ++s)
      if (s < nsrc) {
        f32x4 u0, u1;
        ld2(t0 + s, u0, u1);
        ag[0] = ag[0] + u0;
        ag[1] = ag[1] + u1;
      }
  };

  for (int k = 0; k < a.L; ++k) {
    const lb_persist_layer ly = a.layers[k];
    const bool last = k + 1 == a.L;
    // ================================================================================ edge phase k
    {
      h8 w0h[2][4], w0l[2][4], w1h[2][4], w1l[2][4];
      f32x4 b1v[2], lns[2], lno[2];
      int t = we.q;
      if (has_e) {
        load_idx(t);
        issue_own(t);
      }
      {
        const f32x4* wb = reinterpret_cast<const f32x4*>(ly.we) + lane;
        ms_wload<4, 2>(wb, 2 * w, w0h, w0l);
        ms_wload<4, 2>(wb + 8 * 4 * 2 * 64, 2 * w, w1h, w1l);
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          b1v[c] = reinterpret_cast<const f32x4*>(ly.b1e)[8 * w + 4 * c + g];
          lns[c] = reinterpret_cast<const f32x4*>(ly.lnse)[8 * w + 4 * c + g];
          lno[c] = reinterpret_cast<const f32x4*>(ly.lnoe)[8 * w + 4 * c + g];
        }
      }
      if (k > 0) {  // psr of layer k comes from node phase k-1 (layer 0: the encoder launch before this one)
        if (!ps_grid_barrier(a.bar, ++epoch, n_in_group, &sOk, a.ctrl)) return;
      }
      if (has_e) {
        issue_psr();
        load_idx(min(t + we.stride, we.q_last));
        for (int it = 0; it < we.n_iter; ++it, t += we.stride) {
          f32x4 ve[2], acc[2];
#pragma unroll
          for (int c = 0; c < 2; ++c) {
            ve[c] = nve[c];
            acc[c] = nps[c] + npr[c];
          }
          const int rcur = nr, rb = nrb;
          ps_stage<false>(sB1, w, lane, ve[0], ve[1]);
          {
            const float m = ms_wave_max(guard.see(ve[0], ve[1]));
            if (lane == 0) sMx[0][w] = m;
          }
          issue_own(min(t + we.stride, we.q_last));
          issue_psr();
          load_idx(min(t + 2 * we.stride, we.q_last));
          __syncthreads();
          ms_gemm<4, 2>(sB1, lane, w0h, w0l, acc);
          guard.tile_max(sMx[0]);
          ps_stage<true>(sB2, w, lane, acc[0], acc[1]);
          {
            const float m = ms_wave_max(guard.see(acc[0], acc[1]));
            if (lane == 0) sMx[1][w] = m;
          }
          __syncthreads();
          f32x4 acc2[2] = {b1v[0], b1v[1]};
          ms_gemm<4, 2>(sB2, lane, w1h, w1l, acc2);
          guard.tile_max(sMx[1]);
          {
            const f32x2m p = ms_ln_local(acc2[0], acc2[1]);
            if (g == 0) sRed[n * 4 + w] = p;
          }
          __syncthreads();
          float mean, rs;
          ms_ln_combine(sRed, n, ln_inv_d, ln_pad, mean, rs);
          f32x4 y[2];
#pragma unroll
          for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int j = 0; j < 4; ++j) y[c][j] = (lns[c][j] * rs) * (acc2[c][j] - mean) + lno[c][j];
          asm volatile("" : "+v"(nve[0]), "+v"(nve[1]), "+v"(nps[0]), "+v"(nps[1]), "+v"(npr[0]), "+v"(npr[1]), "+v"(nrb),
                       "+v"(s_i), "+v"(r_i));
          const int row = t * 16 + n;
          const bool valid = row < E;
          if (!last) {
            f32x4* ew = reinterpret_cast<f32x4*>(a.elat) + ((int64_t)t * 8 + 2 * w) * 64 + lane;
            ew[0] = ve[0] + y[0];
            ew[64] = ve[1] + y[1];
          }
          const int rr = valid ? rcur : (-1 - n);
          const int r_prev = __builtin_amdgcn_update_dpp(-2, rr, 0x111, 0xF, 0xF, false);
          const bool head = (n == 0) || (rr != r_prev);
          const unsigned H = (unsigned)(__ballot(head) & 0xffffull);
          const unsigned below = H & ((2u << n) - 1u);
          const int segstart = 31 - __clz(below);
          const bool tail = (n == 15) || ((H >> (n + 1)) & 1u);
          const float m1 = (n >= 1 && segstart <= n - 1) ? 1.f : 0.f, m2 = (n >= 2 && segstart <= n - 2) ? 1.f : 0.f;
          const float m4 = (n >= 4 && segstart <= n - 4) ? 1.f : 0.f, m8 = (n >= 8 && segstart <= n - 8) ? 1.f : 0.f;
          if (!valid) {
            y[0] = zero;
            y[1] = zero;
          }
          lb_scan8(y[0], y[1], m1, m2, m4, m8);
          if (tail && valid) {
            int slot01;
            const bool complete = lb_seg_complete(rb, rr, segstart, n, t, E, slot01);
            if (complete) {
              ps_st(r_agg, (int64_t)rr * 32 + 8 * w + g, y[0]);
              ps_st(r_agg, (int64_t)rr * 32 + 8 * w + 4 + g, y[1]);
            } else {
              ps_st(r_part, ((int64_t)t * 2 + slot01) * 32 + 8 * w + g, y[0]);
              ps_st(r_part, ((int64_t)t * 2 + slot01) * 32 + 8 * w + 4 + g, y[1]);
            }
          }
        }
      }
    }
    // ================================================================================ node phase k
    {
      const bool proj = !last;
      h8 w0h[2][8], w0l[2][8], w1h[2][4], w1l[2][4], wph[4][4], wpl[4][4];
      f32x4 b0v[2], b1v[2], lns[2], lno[2], bpv[4];
      int t = wn.q;
      int64_t rc = 0;
      bool valid = false;
      if (has_n) node_rows_own(t, rc, valid);
      {
        const f32x4* wb = reinterpret_cast<const f32x4*>(ly.wn) + lane;
        ms_wload<8, 2>(wb, 2 * w, w0h, w0l);
        const f32x4* wb1 = wb + 8 * 8 * 2 * 64;
        ms_wload<4, 2>(wb1, 2 * w, w1h, w1l);
        if (proj) ms_wload<4, 4>(wb1 + 8 * 4 * 2 * 64, 4 * w, wph, wpl);
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          b0v[c] = reinterpret_cast<const f32x4*>(ly.b0n)[8 * w + 4 * c + g];
          b1v[c] = reinterpret_cast<const f32x4*>(ly.b1n)[8 * w + 4 * c + g];
          lns[c] = reinterpret_cast<const f32x4*>(ly.lnsn)[8 * w + 4 * c + g];
          lno[c] = reinterpret_cast<const f32x4*>(ly.lnon)[8 * w + 4 * c + g];
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) bpv[c] = proj ? reinterpret_cast<const f32x4*>(ly.bp)[16 * w + 4 * c + g] : zero;
      }
      if (!ps_grid_barrier(a.bar, ++epoch, n_in_group, &sOk, a.ctrl)) return;
      if (has_n) {
        for (int it = 0; it < wn.n_iter; ++it, t += wn.stride) {
          if (it > 0) node_rows_own(t, rc, valid);
          node_agg(rc);
          ps_stage<false>(sB1, w, lane, xa[0], xa[1]);
          ps_stage<false>(sB1, 4 + w, lane, ag[0], ag[1]);
          {
            const float m = ms_wave_max(fmaxf(guard.see(xa[0], xa[1]), guard.see(ag[0], ag[1])));
            if (lane == 0) sMx[0][w] = m;
          }
          __syncthreads();
          f32x4 acc[2] = {b0v[0], b0v[1]};
          ms_gemm<8, 2>(sB1, lane, w0h, w0l, acc);
          guard.tile_max(sMx[0]);
          ps_stage<true>(sB2, w, lane, acc[0], acc[1]);
          {
            const float m = ms_wave_max(guard.see(acc[0], acc[1]));
            if (lane == 0) sMx[1][w] = m;
          }
          __syncthreads();
          f32x4 acc2[2] = {b1v[0], b1v[1]};
          ms_gemm<4, 2>(sB2, lane, w1h, w1l, acc2);
          guard.tile_max(sMx[1]);
          {
            const f32x2m p = ms_ln_local(acc2[0], acc2[1]);
            if (g == 0) sRed[n * 4 + w] = p;
          }
          __syncthreads();
          float mean, rs;
          ms_ln_combine(sRed, n, ln_inv_d, ln_pad, mean, rs);
          f32x4 y[2];
#pragma unroll
          for (int c = 0; c < 2; ++c) {
#pragma unroll
            for (int j = 0; j < 4; ++j) y[c][j] = (lns[c][j] * rs) * (acc2[c][j] - mean) + lno[c][j];
            y[c] = xa[c] + y[c];  // residual (gns.py:120-122)
            if (valid) reinterpret_cast<f32x4*>(a.nlat)[rc * 32 + 8 * w + 4 * c + g] = y[c];
          }
          if (proj) {
            ps_stage<false>(sB3, w, lane, y[0], y[1]);
            {
              const float m = ms_wave_max(guard.see(y[0], y[1]));
              if (lane == 0) sMx[2][w] = m;
            }
            __syncthreads();
            guard.tile_max(sMx[2]);
            f32x4 accp[4] = {bpv[0], bpv[1], bpv[2], bpv[3]};
            ms_gemm<4, 4>(sB3, lane, wph, wpl, accp);
            if (valid) {
#pragma unroll
              for (int c = 0; c < 4; ++c) ps_st(r_psr, rc * 64 + 16 * w + 4 * c + g, accp[c]);
            }
          }
        }
      }
    }
  }
  guard.commit(a.ctrl, lane);
}

// ---------------------------------------------------------------------------------------------------- host
int lbk_gns_persist(lb_engine* e, const lb_persist_args& a_in) {
  lb_persist_args a = a_in;
  LB_HIP(hipMemsetAsync(a.bar, 0, sizeof(unsigned) * PS_BAR_STRIDE * 12, e->stream));
  LB_LAUNCH_TIMED(e, k_gns_persist, dim3(a.grid), dim3(MS_THREADS), a);
  LB_HIP(hipGetLastError());
  static const bool dbg = getenv("LB_PERSIST_DBG") && getenv("LB_PERSIST_DBG")[0] == '1';
  if (dbg) {  // debug: barrier words after the launch
    unsigned h[PS_BAR_STRIDE * 12];
    const hipError_t se = hipDeviceSynchronize();
    (void)hipMemcpy(h, a.bar, sizeof(h), hipMemcpyDeviceToHost);
    fprintf(stderr, "[persist] sync=%s grid=%d L=%d groups:", hipGetErrorString(se), a.grid, a.L);
    for (int i = 0; i < 8; ++i) fprintf(stderr, " %u", h[i * PS_BAR_STRIDE]);
    fprintf(stderr, " top=%u gen=%u err=%u\n", h[8 * PS_BAR_STRIDE], h[9 * PS_BAR_STRIDE], h[10 * PS_BAR_STRIDE]);
  }
  return LB_OK;
}
