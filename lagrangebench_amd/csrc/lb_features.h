// lb_features.h - per-particle bodies of the node-feature assembly and of the integrator, shared by the stand-alone
// kernels (lb_state.hip) and by the launches they ride along with in a rollout step: the feature rows are written by
// extra workgroups of the neighbor-search launch (both only read the position window), the integrator runs in the
// decoder's epilogue (round 3: two launches fewer per step).
#pragma once
#include "lb_device.h"

// -------------------------------------------------------------------------- node features
// One thread per particle.  Writes the fp32 network input row [vel_hist | vel_mag | bound | force |
// embedding | 0-pad] (gns.py:135-169 column order) and, when asked, the fp64 feature arrays the
// Python FeatureDict exposes.
__device__ __forceinline__ void lb_node_features_body(const lb_geom& g, int64_t BN, const double* __restrict__ win,
                                                      int step, const int32_t* __restrict__ ptype,
                                                      const double* __restrict__ force_buf, float* __restrict__ xnode,
                                                      const float* __restrict__ embed, int emb, int ntypes,
                                                      double* __restrict__ vel_hist, double* __restrict__ vel_mag,
                                                      double* __restrict__ bound, double* __restrict__ force_out,
                                                      int64_t gi) {
  const int K = g.isl - 1, dim = g.dim;
  float* x = xnode ? xnode + gi * g.kpad : nullptr;
  int col = 0;
  double pprev[3], pcur[3] = {0, 0, 0};
  for (int d = 0; d < dim; ++d) pprev[d] = lb_pos(win, g, BN, step, 0, d, gi);
  for (int t = 0; t < K; ++t) {
    double s2 = 0.0;
    for (int d = 0; d < dim; ++d) {
      pcur[d] = lb_pos(win, g, BN, step, t + 1, d, gi);
      const double v = lb_disp1(pcur[d], pprev[d], g.box[d], g.half_box[d], g.periodic, g.f32);
      const double nv = lb_r(lb_r(v - g.vel_mean[d], g.f32) / g.vel_std[d], g.f32);
      if (x) x[t * dim + d] = (float)nv;
      if (vel_hist) vel_hist[gi * (K * dim) + t * dim + d] = nv;
      s2 = (d == 0) ? lb_r(nv * nv, g.f32) : lb_r(s2 + lb_r(nv * nv, g.f32), g.f32);
      pprev[d] = pcur[d];
    }
    if (g.has_vel_mag) {
      const double m = lb_r(sqrt(s2), g.f32);
      if (x) x[K * dim + t] = (float)m;
      if (vel_mag) vel_mag[gi * K + t] = m;
    }
  }
  col = K * dim + (g.has_vel_mag ? K : 0);
  if (K == 0)
    for (int d = 0; d < dim; ++d) pcur[d] = lb_pos(win, g, BN, step, 0, d, gi);
  if (g.has_bound) {
    for (int d = 0; d < dim; ++d) {
      double lo = lb_r(lb_r(pcur[d] - g.bound_lo[d], g.f32) / g.rc, g.f32);
      double hi = lb_r(lb_r(g.bound_hi[d] - pcur[d], g.f32) / g.rc, g.f32);
      lo = fmin(fmax(lo, -1.0), 1.0);
      hi = fmin(fmax(hi, -1.0), 1.0);
      if (x) {
        x[col + d] = (float)lo;
        x[col + dim + d] = (float)hi;
      }
      if (bound) {
        bound[gi * 2 * dim + d] = lo;
        bound[gi * 2 * dim + dim + d] = hi;
      }
    }
    col += 2 * dim;
  }
  if (g.force_kind != LB_FORCE_NONE) {
    for (int d = 0; d < dim; ++d) {
      double f;
      if (g.force_kind == LB_FORCE_PIECEWISE)
        f = (pcur[g.force_axis] > g.force_split) ? g.force_hi[d] : g.force_lo[d];
      else
        f = force_buf[gi * dim + d];
      if (x) x[col + d] = (float)f;
      if (force_out) force_out[gi * dim + d] = f;
    }
    col += dim;
  }
  if (x) {
    if (ntypes > 1) {
      int t = ptype[gi];
      if (t < 0) t += ntypes;  // jnp negative index wraps (PAD_VALUE = -1 -> last row)
      t = t < 0 ? 0 : (t >= ntypes ? ntypes - 1 : t);
      for (int j = 0; j < emb; ++j) x[col + j] = embed[t * emb + j];
      col += emb;
    }
    for (int j = col; j < g.kpad; ++j) x[j] = 0.f;
  }
}

// One column of particle gi's network input row - the same operations on the same values as lb_node_features_body,
// arranged so that the 64 lanes of a wave produce 64 columns of ONE row: a search wave writes the row of its
// receiver with one coalesced store per 64 columns (a rollout step then has no feature launch at all).
__device__ __forceinline__ float lb_node_feature_column(const lb_geom& g, int64_t BN, const double* __restrict__ win,
                                                        int step, const lb_feat_job& f, int64_t gi, int col) {
  const int K = g.isl - 1, dim = g.dim;
  const int c_mag = K * dim, c_bnd = c_mag + (g.has_vel_mag ? K : 0), c_frc = c_bnd + (g.has_bound ? 2 * dim : 0);
  const int c_emb = c_frc + (g.force_kind != LB_FORCE_NONE ? dim : 0), c_end = c_emb + (f.ntypes > 1 ? f.emb : 0);
  auto nvel = [&](int t, int d) -> double {
    const double p1 = lb_pos(win, g, BN, step, t + 1, d, gi), p0 = lb_pos(win, g, BN, step, t, d, gi);
    const double v = lb_disp1(p1, p0, g.box[d], g.half_box[d], g.periodic, g.f32);
    return lb_r(lb_r(v - g.vel_mean[d], g.f32) / g.vel_std[d], g.f32);
  };
  if (col < c_mag) return (float)nvel(col / dim, col % dim);
  if (col < c_bnd) {
    const int t = col - c_mag;
    double s2 = 0.0;
    for (int d = 0; d < dim; ++d) {
      const double nv = nvel(t, d);
      s2 = (d == 0) ? lb_r(nv * nv, g.f32) : lb_r(s2 + lb_r(nv * nv, g.f32), g.f32);
    }
    return (float)lb_r(sqrt(s2), g.f32);
  }
  if (col < c_frc) {
    const int k = col - c_bnd, d = k % dim;
    const double pc = lb_pos(win, g, BN, step, K, d, gi);
    double r = k < dim ? lb_r(lb_r(pc - g.bound_lo[d], g.f32) / g.rc, g.f32) : lb_r(lb_r(g.bound_hi[d] - pc, g.f32) / g.rc, g.f32);
    return (float)fmin(fmax(r, -1.0), 1.0);
  }
  if (col < c_emb) {
    const int d = col - c_frc;
    if (g.force_kind == LB_FORCE_PIECEWISE)
      return (float)((lb_pos(win, g, BN, step, K, g.force_axis, gi) > g.force_split) ? g.force_hi[d] : g.force_lo[d]);
    return (float)f.force[gi * dim + d];
  }
  if (col < c_end) {
    int t = f.ptype[gi];
    if (t < 0) t += f.ntypes;  // jnp negative index wraps (PAD_VALUE = -1 -> last row)
    t = t < 0 ? 0 : (t >= f.ntypes ? f.ntypes - 1 : t);
    return f.embed[t * f.emb + (col - c_emb)];
  }
  return 0.f;
}
// the rows of `count` particles (ids[0 .. count), e.g. the own particles of a cell) by one wave: (particle, column)
// pairs are dealt out to the lanes four rounds at a time, so that the position loads of up to 256 columns are in flight
// together (one row at a time cost the per-cell search kernel a memory round trip per particle)
template <typename IDS>
__device__ __forceinline__ void lb_node_features_wave_multi(const lb_geom& g, int64_t BN, const double* __restrict__ win,
                                                            int step, const lb_feat_job& f, IDS ids, int count) {
  const int lane = threadIdx.x & 63, kp = f.kpad, total = count * kp;
  for (int base = 0; base < total; base += 256) {
    float v[4];
    int64_t dst[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int idx = base + 64 * u + lane;
      const bool ok = idx < total;
      const int p = ok ? idx / kp : 0, col = ok ? idx - p * kp : 0;
      const int64_t gi = ids(p);
      dst[u] = ok ? gi * kp + col : -1;
      v[u] = lb_node_feature_column(g, BN, win, step, f, gi, col);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (dst[u] >= 0) f.xnode[dst[u]] = v[u];
  }
}
__device__ __forceinline__ void lb_node_features_wave(const lb_geom& g, int64_t BN, const double* __restrict__ win,
                                                      int step, const lb_feat_job& f, int64_t gi) {
  float* const x = f.xnode + gi * f.kpad;
  for (int c0 = 0; c0 < f.kpad; c0 += 64) {
    const int col = c0 + (threadIdx.x & 63);
    if (col < f.kpad) x[col] = lb_node_feature_column(g, BN, win, step, f, gi, col);
  }
}

// integrate_fn + kinematic select + window advance + prediction store for particle gi (case.py:230-259,
// rollout.py:61-73,165-167); `accv` = the network's normalised acceleration of this particle
// what the integrator reads of particle gi, fetched ahead of the arithmetic that produces its acceleration
struct lb_integ_in {  // (named components, selected with compares: an array indexed by the run-time d lands in scratch)
  double p1x, p1y, p1z, p0x, p0y, p0z;
  int pt;
};
__device__ __forceinline__ double lb_sel3(int d, double x, double y, double z) { return d == 0 ? x : (d == 1 ? y : z); }
__device__ __forceinline__ lb_integ_in lb_integrate_fetch(const lb_geom& g, int64_t BN, const double* __restrict__ win,
                                                          int step, const int32_t* __restrict__ ptype, int64_t gi) {
  lb_integ_in in;
  in.pt = ptype[gi];
  in.p1x = lb_pos(win, g, BN, step, g.isl - 1, 0, gi);
  in.p0x = lb_pos(win, g, BN, step, g.isl - 2, 0, gi);
  in.p1y = lb_pos(win, g, BN, step, g.isl - 1, 1, gi);  // (dim >= 2)
  in.p0y = lb_pos(win, g, BN, step, g.isl - 2, 1, gi);
  in.p1z = g.dim == 3 ? lb_pos(win, g, BN, step, g.isl - 1, 2, gi) : 0.0;
  in.p0z = g.dim == 3 ? lb_pos(win, g, BN, step, g.isl - 2, 2, gi) : 0.0;
  return in;
}
__device__ __forceinline__ void lb_integrate_body(const lb_geom& g, int64_t BN, double* __restrict__ win, int step,
                                                  const int32_t* __restrict__ ptype, const float* accv,
                                                  const double* __restrict__ target, const double* __restrict__ traj,
                                                  int T, double* __restrict__ pred, int pred_T, int64_t gi,
                                                  const lb_integ_in* pre = nullptr) {
  const int b = (int)(gi / g.N), i = (int)(gi % g.N);
  const int pt = pre ? pre->pt : ptype[gi];
  const bool kinematic = (pt == 1) || (pt == 2) || (pt == -1);  // utils.py:28-35
  const int slot_new = (step + g.isl) % g.isl;
  int tf = g.isl + step;
  if (tf > T - 1) tf = T - 1;  // JAX clamps the out-of-range gather (rollout.py:159)
  for (int d = 0; d < g.dim; ++d) {
    double out;
    if (kinematic) {
      out = target ? target[gi * g.dim + d] : traj[(gi * T + tf) * g.dim + d];
    } else {
      const double p1 = pre ? lb_sel3(d, pre->p1x, pre->p1y, pre->p1z) : lb_pos(win, g, BN, step, g.isl - 1, d, gi);
      const double p0 = pre ? lb_sel3(d, pre->p0x, pre->p0y, pre->p0z) : lb_pos(win, g, BN, step, g.isl - 2, d, gi);
      const double a = lb_r(g.acc_mean[d] + lb_r((double)accv[d] * g.acc_std[d], g.f32), g.f32);
      const double v = lb_disp1(p1, p0, g.box[d], g.half_box[d], g.periodic, g.f32);
      out = lb_shift1(p1, lb_r(v + a, g.f32), g.box[d], g.periodic, g.f32);
    }
    win[((int64_t)slot_new * g.dim + d) * BN + gi] = out;
    if (pred && step < pred_T) pred[(((int64_t)b * pred_T + step) * g.N + i) * g.dim + d] = out;
  }
}

