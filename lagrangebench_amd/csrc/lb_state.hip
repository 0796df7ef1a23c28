// lb_state.hip - position-window SoA ring, node features, integrator, rollout metrics.
//
// Reference functions replaced (paths relative to the reference repo):
//   feature_transform, node part      lagrangebench/case_setup/features.py:47-107
//   hk.Embed lookup + concat          lagrangebench/models/gns.py:164-169
//   integrate_fn                      lagrangebench/case_setup/case.py:230-259
//   _forward_eval (mask + shift)      lagrangebench/evaluate/rollout.py:61-73
//   predictions.at[:, step].set       lagrangebench/evaluate/rollout.py:165-167
//   MetricsComputer.mse / .mae        lagrangebench/evaluate/metrics.py:139-147
//
// HBM layout: win[slot][d][b*N+i] fp64 - structure-of-arrays so that every kernel reads positions
// with unit stride across lanes; the window is a ring of isl slots (frame f of the window that
// starts at step s lives in slot (s+f) % isl), so advancing the window writes ONE frame instead
// of the reference's concatenate-and-copy of the whole (N, isl, dim) array.
#include "lb_features.h"

__global__ void k_load_window(lb_geom g, int64_t BN, const double* __restrict__ traj, int T, int t0,
                              int step, double* __restrict__ win, lb_ctrl* __restrict__ ctrl) {
  int64_t gi = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gi == 0) ctrl->step = step;
  if (gi >= BN) return;
  for (int f = 0; f < g.isl; ++f) {
    const int slot = (step + f) % g.isl;
    for (int d = 0; d < g.dim; ++d)
      win[((int64_t)slot * g.dim + d) * BN + gi] = traj[(gi * T + (t0 + f)) * g.dim + d];
  }
}

// Window of rollout step `step` rebuilt from the input frames and the predictions made so far: frame f of the sequence
// [traj[:, :isl] | pred[:, 0], pred[:, 1], ...] (pred is (B, pred_T, N, dim), the layout k_integrate writes).
__global__ void k_load_window_resume(lb_geom g, int64_t BN, const double* __restrict__ traj, int T,
                                     const double* __restrict__ pred, int pred_T, int step, double* __restrict__ win,
                                     lb_ctrl* __restrict__ ctrl) {
  int64_t gi = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gi == 0) ctrl->step = step;
  if (gi >= BN) return;
  const int b = (int)(gi / g.N), i = (int)(gi % g.N);
  for (int f = 0; f < g.isl; ++f) {
    const int slot = (step + f) % g.isl, frame = step + f;
    for (int d = 0; d < g.dim; ++d)
      win[((int64_t)slot * g.dim + d) * BN + gi] =
          frame < g.isl ? traj[(gi * T + frame) * g.dim + d]
                        : pred[(((int64_t)b * pred_T + (frame - g.isl)) * g.N + i) * g.dim + d];
  }
}
int lbk_load_window_resume(lb_engine* e, const double* traj, int T, const double* pred, int pred_T, int step) {
  const int nb = (int)((e->BN + 255) / 256);
  hipLaunchKernelGGL(k_load_window_resume, dim3(nb), dim3(256), 0, e->stream, e->g, e->BN, traj, T, pred, pred_T, step,
                     e->win, e->ctrl);
  LB_HIP(hipGetLastError());
  return LB_OK;
}

int lbk_load_window(lb_engine* e, const double* traj, int T, int t0, int step) {
  const int nb = (int)((e->BN + 255) / 256);
  hipLaunchKernelGGL(k_load_window, dim3(nb), dim3(256), 0, e->stream, e->g, e->BN, traj, T, t0,
                     step, e->win, e->ctrl);
  LB_HIP(hipGetLastError());
  return LB_OK;
}

__global__ void k_read_window(lb_geom g, int64_t BN, const double* __restrict__ win,
                              const lb_ctrl* __restrict__ ctrl, double* __restrict__ out) {
  int64_t gi = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gi >= BN) return;
  const int step = ctrl->step;
  for (int f = 0; f < g.isl; ++f)
    for (int d = 0; d < g.dim; ++d)
      out[(gi * g.isl + f) * g.dim + d] = lb_pos(win, g, BN, step, f, d, gi);
}

int lbk_read_window(lb_engine* e, double* out) {
  const int nb = (int)((e->BN + 255) / 256);
  hipLaunchKernelGGL(k_read_window, dim3(nb), dim3(256), 0, e->stream, e->g, e->BN, e->win, e->ctrl,
                     out);
  LB_HIP(hipGetLastError());
  return LB_OK;
}

// -------------------------------------------------------------------------- node features
// One thread per particle (body: lb_features.h).  Writes the fp32 network input row [vel_hist | vel_mag | bound |
// force | embedding | 0-pad] (gns.py:135-169 column order) and, when asked, the fp64 feature arrays the Python
// FeatureDict exposes.
__global__ void k_node_features(lb_geom g, int64_t BN, const double* __restrict__ win,
                                const lb_ctrl* __restrict__ ctrl, const int32_t* __restrict__ ptype,
                                const double* __restrict__ force_buf, float* __restrict__ xnode,
                                const float* __restrict__ embed, int emb, int ntypes,
                                double* __restrict__ vel_hist, double* __restrict__ vel_mag,
                                double* __restrict__ bound, double* __restrict__ force_out) {
  if (xnode && ctrl->overflow_step >= 0) return;
  int64_t gi = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gi >= BN) return;
  lb_node_features_body(g, BN, win, ctrl->step, ptype, force_buf, xnode, embed, emb, ntypes, vel_hist, vel_mag, bound,
                        force_out, gi);
}

int lbk_node_features(lb_engine* e, float* xnode, const float* embed, int emb, int ntypes,
                      double* vel_hist, double* vel_mag, double* bound, double* force) {
  const int nb = (int)((e->BN + 255) / 256);
  hipLaunchKernelGGL(k_node_features, dim3(nb), dim3(256), 0, e->stream, e->g, e->BN, e->win,
                     e->ctrl, e->ptype, e->force, xnode, embed, emb, ntypes, vel_hist, vel_mag,
                     bound, force);
  LB_HIP(hipGetLastError());
  return LB_OK;
}

// Same feature row without embedding columns, at a caller-chosen row stride (SEGNN input).
int lbk_node_features_raw(lb_engine* e, float* xnode, int kpad) {
  lb_geom g = e->g;
  g.kpad = kpad;
  const int nb = (int)((e->BN + 255) / 256);
  hipLaunchKernelGGL(k_node_features, dim3(nb), dim3(256), 0, e->stream, g, e->BN, e->win, e->ctrl,
                     e->ptype, e->force, xnode, nullptr, 0, 1, nullptr, nullptr, nullptr, nullptr);
  LB_HIP(hipGetLastError());
  return LB_OK;
}

// ------------------------------------------------------------------------------ integrator
// The step counter is advanced by the LAST workgroup to finish (every workgroup has read ctrl->step by
// then): no separate one-thread launch at the end of every step.
__global__ void k_integrate(lb_geom g, int64_t BN, double* __restrict__ win,
                            lb_ctrl* __restrict__ ctrl, int32_t* __restrict__ blocks_done,
                            const int32_t* __restrict__ ptype,
                            const float* __restrict__ acc, int acc_stride,
                            const double* __restrict__ target, const double* __restrict__ traj,
                            int T, double* __restrict__ pred, int pred_T) {
  if (ctrl->overflow_step >= 0) return;
  int64_t gi = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int step = ctrl->step;
  __syncthreads();  // every thread of this workgroup holds `step` before the workgroup can report completion
  if (threadIdx.x == 0) {
    __threadfence();
    if (atomicAdd(blocks_done, 1) == (int)gridDim.x - 1) {
      *blocks_done = 0;
      ctrl->step = step + 1;
    }
  }
  if (gi >= BN) return;
  lb_integrate_body(g, BN, win, step, ptype, acc + gi * acc_stride, target, traj, T, pred, pred_T, gi);
}

// case.integrate alone (stateless), case.py:230-259.
__global__ void k_case_integrate(lb_geom g, int64_t BN, int mode, const float* __restrict__ pred,
                                 const double* __restrict__ pos_seq, int T,
                                 double* __restrict__ out) {
  int64_t gi = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gi >= BN) return;
  for (int d = 0; d < g.dim; ++d) {
    const double p1 = pos_seq[(gi * T + (T - 1)) * g.dim + d];
    double nv;
    if (mode == 1) {
      nv = lb_r(g.vel_mean[d] + lb_r((double)pred[gi * g.dim + d] * g.vel_std[d], g.f32), g.f32);
    } else {
      const double p0 = pos_seq[(gi * T + (T - 2)) * g.dim + d];
      const double a = lb_r(g.acc_mean[d] + lb_r((double)pred[gi * g.dim + d] * g.acc_std[d], g.f32), g.f32);
      nv = lb_r(lb_disp1(p1, p0, g.box[d], g.half_box[d], g.periodic, g.f32) + a, g.f32);
    }
    out[gi * g.dim + d] = lb_shift1(p1, nv, g.box[d], g.periodic, g.f32);
  }
}

int lbk_case_integrate(lb_engine* e, int mode, const float* pred, const double* pos_seq, int T,
                       double* out) {
  const int nb = (int)((e->BN + 255) / 256);
  hipLaunchKernelGGL(k_case_integrate, dim3(nb), dim3(256), 0, e->stream, e->g, e->BN, mode, pred,
                     pos_seq, T, out);
  LB_HIP(hipGetLastError());
  return LB_OK;
}

int lbk_integrate(lb_engine* e, const float* acc, int acc_stride, const double* target,
                  const double* traj, int T, double* pred, int pred_T) {
  const int nb = (int)((e->BN + 255) / 256);
  lb_tic(e, LB_T_INTEGRATE);
  hipLaunchKernelGGL(k_integrate, dim3(nb), dim3(256), 0, e->stream, e->g, e->BN, e->win, e->ctrl,
                     e->blocks_done, e->ptype, acc, acc_stride, target, traj, T, pred, pred_T);
  lb_toc(e);
  LB_HIP(hipGetLastError());
  return LB_OK;
}

// --------------------------------------------------------------------------------- metrics
// One workgroup per (trajectory, step): fixed-order strided accumulation + LDS tree, fp64.
__global__ void __launch_bounds__(256)
    k_metrics(lb_geom g, const double* __restrict__ pred, int pred_T,
              const double* __restrict__ target, int target_T, int n_steps,
              double* __restrict__ mse, double* __restrict__ mae) {
  __shared__ double s2[256], s1[256];
  const int t = blockIdx.x, b = blockIdx.y;
  double a2 = 0.0, a1 = 0.0;
  const int n = g.N * g.dim;
  for (int k = threadIdx.x; k < n; k += 256) {
    const int i = k / g.dim, d = k % g.dim;
    const double p = pred[(((int64_t)b * pred_T + t) * g.N + i) * g.dim + d];
    const double q = target[(((int64_t)b * target_T + t) * g.N + i) * g.dim + d];
    const double dd = lb_disp1(p, q, g.box[d], g.half_box[d], g.periodic);
    a2 += dd * dd;
    a1 += fabs(dd);
  }
  s2[threadIdx.x] = a2;
  s1[threadIdx.x] = a1;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (threadIdx.x < off) {
      s2[threadIdx.x] += s2[threadIdx.x + off];
      s1[threadIdx.x] += s1[threadIdx.x + off];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    if (mse) mse[(int64_t)b * n_steps + t] = s2[0] / (double)n;
    if (mae) mae[(int64_t)b * n_steps + t] = s1[0] / (double)n;
  }
}

// MetricsComputer "e_kin" (evaluate/metrics.py:98-125,157-160): kinetic energy of strided frames,
// E[k] = dx^dim * sum_{i,d} (disp(x[1 + k*stride], x[k*stride]) / dt)^2, one workgroup per (traj, k).
__global__ void __launch_bounds__(256)
    k_ekin(lb_geom g, const double* __restrict__ roll, int T, int stride, int n_out, double inv_dt,
           double vol, double* __restrict__ out) {
  __shared__ double s2[256];
  const int k = blockIdx.x, b = blockIdx.y;
  const int t0 = k * stride, t1 = 1 + k * stride;
  double a2 = 0.0;
  const int n = g.N * g.dim;
  for (int q = threadIdx.x; q < n; q += 256) {
    const int i = q / g.dim, d = q % g.dim;
    const double p1 = roll[(((int64_t)b * T + t1) * g.N + i) * g.dim + d];
    const double p0 = roll[(((int64_t)b * T + t0) * g.N + i) * g.dim + d];
    const double v = lb_disp1(p1, p0, g.box[d], g.half_box[d], g.periodic) * inv_dt;
    a2 += v * v;
  }
  s2[threadIdx.x] = a2;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (threadIdx.x < off) s2[threadIdx.x] += s2[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[(int64_t)b * n_out + k] = s2[0] * vol;
}

int lbk_ekin(lb_engine* e, const double* roll, int T, int stride, int n_out, double dt, double dx,
             double* out) {
  if (n_out <= 0) return LB_OK;
  double vol = 1.0;
  for (int d = 0; d < e->g.dim; ++d) vol *= dx;
  hipLaunchKernelGGL(k_ekin, dim3(n_out, e->g.B), dim3(256), 0, e->stream, e->g, roll, T, stride, n_out,
                     1.0 / dt, vol, out);
  LB_HIP(hipGetLastError());
  return LB_OK;
}

int lbk_metrics(lb_engine* e, const double* pred, int pred_T, const double* target, int target_T,
                int n_steps, double* mse, double* mae) {
  if (n_steps <= 0) return LB_OK;
  hipLaunchKernelGGL(k_metrics, dim3(n_steps, e->g.B), dim3(256), 0, e->stream, e->g, pred, pred_T,
                     target, target_T, n_steps, mse, mae);
  LB_HIP(hipGetLastError());
  return LB_OK;
}
