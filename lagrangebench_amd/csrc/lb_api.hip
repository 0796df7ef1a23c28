// lb_api.hip - the C ABI of liblbhip.so (include/lbhip.h): engine object, buffers, weight
// packing, the device-resident rollout driver and the HIP-event timers used by bench.py.
#include <cstddef>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <cmath>

#include <initializer_list>
#include <tuple>

#include "lb_internal.h"
#include "lb_msplit.h"

thread_local std::string g_lb_err;

int lb_fail(int code, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_lb_err = buf;
  return code;
}

extern "C" const char* lb_strerror(int code) {
  switch (code) {
    case LB_OK: return "ok";
    case LB_ERR_ARG: return "bad argument";
    case LB_ERR_HIP: return "HIP runtime error";
    case LB_ERR_STATE: return "call order violated";
    case LB_ERR_DENSITY: return "cell stencil / row exceeds LDS tile bounds";
    case LB_ERR_UNSUPPORTED: return "not built yet";
    default: return "unknown";
  }
}
extern "C" const char* lb_last_error(void) { return g_lb_err.c_str(); }
extern "C" int lb_version(void) { return 100; }

// ---------------------------------------------------------------------------------- timers
static const char* k_timer_names[LB_T_COUNT] = {
    "cells", "neighbors", "node_features", "enc_node", "enc_edge", "edge_mlp",
    "aggregate", "node_mlp", "decoder", "integrate", "misc", "processor", "edge_mlp_last"};

static hipEvent_t lb_get_event(lb_engine* e) {
  if (!e->epool.empty()) {
    hipEvent_t ev = e->epool.back();
    e->epool.pop_back();
    return ev;
  }
  hipEvent_t ev;
  (void)hipEventCreate(&ev);
  return ev;
}

void lb_tic(lb_engine* e, int cls) {
  if (!e->timers_on) return;
  lb_timer_rec r;
  r.a = lb_get_event(e);
  r.b = lb_get_event(e);
  r.cls = cls;
  (void)hipEventRecord(r.a, e->stream);
  e->trecs.push_back(r);
}

void lb_tic_single(lb_engine* e, int cls) {
  lb_tic(e, cls);
  e->ext_armed = e->timers_on;
  e->ext_used = false;
}

void lb_toc(lb_engine* e) {
  if (!e->timers_on || e->trecs.empty()) return;
  if (!(e->ext_armed && e->ext_used)) (void)hipEventRecord(e->trecs.back().b, e->stream);
  e->ext_armed = e->ext_used = false;
}

static void lb_timers_collect(lb_engine* e) {
  if (e->trecs.empty()) return;
  (void)hipStreamSynchronize(e->stream);
  for (auto& r : e->trecs) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
      e->t_ms[r.cls] += ms;
      e->t_n[r.cls] += 1;
    }
    e->epool.push_back(r.a);
    e->epool.push_back(r.b);
  }
  e->trecs.clear();
}

extern "C" int lb_timers_enable(lb_engine* e, int32_t on) {
  if (!e) return lb_fail(LB_ERR_ARG, "null engine");
  lb_timers_collect(e);
  e->timers_on = on != 0;
  return LB_OK;
}
extern "C" int lb_timers_reset(lb_engine* e) {
  if (!e) return lb_fail(LB_ERR_ARG, "null engine");
  lb_timers_collect(e);
  for (int i = 0; i < LB_T_COUNT; ++i) {
    e->t_ms[i] = 0;
    e->t_n[i] = 0;
  }
  return LB_OK;
}
extern "C" int32_t lb_timer_count(void) { return LB_T_COUNT; }
extern "C" const char* lb_timer_name(int32_t cls) {
  return (cls >= 0 && cls < LB_T_COUNT) ? k_timer_names[cls] : "";
}
extern "C" int lb_timer_get(lb_engine* e, int32_t cls, double* ms_out, int64_t* launches_out) {
  if (!e || cls < 0 || cls >= LB_T_COUNT) return lb_fail(LB_ERR_ARG, "bad timer class");
  lb_timers_collect(e);
  if (ms_out) *ms_out = e->t_ms[cls];
  if (launches_out) *launches_out = e->t_n[cls];
  return LB_OK;
}

// ---------------------------------------------------------------------------------- engine

extern "C" int lb_engine_create(const lb_case_desc* d, void* hip_stream, lb_engine** out) {
  if (!d || !out) return lb_fail(LB_ERR_ARG, "null argument");
  if (d->dim != 2 && d->dim != 3) return lb_fail(LB_ERR_ARG, "dim must be 2 or 3 (got %d)", d->dim);
  if (d->n_particles < 1 || d->batch < 1 || d->isl < 2)
    return lb_fail(LB_ERR_ARG, "need n_particles>=1, batch>=1, isl>=2");
  if (!(d->r_cutoff > 0)) return lb_fail(LB_ERR_ARG, "r_cutoff must be > 0");
  if ((int64_t)d->n_particles * d->batch > (int64_t)1 << 30)
    return lb_fail(LB_ERR_ARG, "B*N too large for int32 node ids");
  lb_engine* e = new lb_engine();
  e->desc = *d;
  e->stream = (hipStream_t)hip_stream;
  e->BN = (int64_t)d->n_particles * d->batch;
  e->timers_on = false;
  for (int i = 0; i < LB_T_COUNT; ++i) e->t_ms[i] = 0, e->t_n[i] = 0;
  {
    const char* f = getenv("LB_FUSED_AGG");
    e->fused_agg = (f && f[0] == '0') ? 0 : 1;
    const char* m = getenv("LB_MATH");
    e->f16x2 = (m && !strcmp(m, "f32")) ? 0 : 1;
    e->math_auto = m ? 0 : 1;  // LB_MATH given: that arithmetic, no guard-driven switch
    const char* gf = getenv("LB_GUARD");
    e->guard_full = (gf && !strcmp(gf, "full")) ? 1 : 0;
    e->guard_sampled = (gf && !strcmp(gf, "sampled")) ? 1 : 0;
  }
  lb_geom& g = e->g;
  memset(&g, 0, sizeof(g));
  g.dim = d->dim;
  g.N = d->n_particles;
  g.B = d->batch;
  g.isl = d->isl;
  g.periodic = d->periodic != 0;
  g.has_bound = d->has_bound != 0;
  g.has_vel_mag = d->has_vel_mag != 0;
  g.force_kind = d->force_kind;
  g.force_axis = d->force_axis;
  g.f32 = d->geometry_f32 != 0;
  // dtype=float32: every constant is the float the reference would hold (cutoff^2 is squared in Python floats and
  // THEN cast: jax-md prunes with position.dtype.type(cutoff_sq))
  auto R = [&](double x) { return g.f32 ? (double)(float)x : x; };
  g.rc = R(d->r_cutoff);
  g.rc2 = R(d->r_cutoff * d->r_cutoff);
  // jax-md partition.py: box and cell_size are float32; cell list only if cutoff < box/3.
  bool use_cells = true;
  for (int k = 0; k < d->dim; ++k) {
    const float b32 = (float)d->box[k];
    if (!((float)d->r_cutoff < b32 / 3.0f)) use_cells = false;
  }
  g.use_cell_list = use_cells;
  g.ncells = 1;
  for (int k = 0; k < 3; ++k) {
    g.ncell[k] = 1;
    g.cell_size[k] = 1.0;
    g.box[k] = k < d->dim ? R(d->box[k]) : 1.0;
    g.half_box[k] = g.box[k] * 0.5;
    g.vel_mean[k] = R(d->vel_mean[k]);
    g.vel_std[k] = R(d->vel_std[k]);
    g.acc_mean[k] = R(d->acc_mean[k]);
    g.acc_std[k] = R(d->acc_std[k]);
    g.bound_lo[k] = R(d->bound_lo[k]);
    g.bound_hi[k] = R(d->bound_hi[k]);
    g.force_lo[k] = R(d->force_lo[k]);  // dtype=float32: external_force_fn(position) is evaluated in float32 (features.py:105-107)
    g.force_hi[k] = R(d->force_hi[k]);
  }
  g.force_split = R(d->force_split);
  if (use_cells) {
    for (int k = 0; k < d->dim; ++k) {
      const float b32 = (float)d->box[k];
      const float cps = floorf(b32 / (float)d->r_cutoff);
      g.ncell[k] = (int)cps;
      g.cell_size[k] = (double)(b32 / cps);
      g.ncells *= g.ncell[k];
    }
    g.nstencil = d->dim == 2 ? 9 : 27;
  } else {
    g.nstencil = 1;
    // all-pairs candidates: one "cell" per trajectory - staged whole in LDS up to LB_MAX_STENCIL_CAND particles, beyond
    // that the wave-per-receiver kernel walks it from global memory (dense fall-back)
    if (d->n_particles > LB_MAX_STENCIL_CAND) e->nl_dense = true;
    for (int k = 0; k < d->dim; ++k) g.cell_size[k] = 1e300;  // every particle -> cell 0
  }
  const int K = d->isl - 1;
  g.node_in = K * d->dim + (g.has_vel_mag ? K : 0) + (g.has_bound ? 2 * d->dim : 0) +
              (g.force_kind != LB_FORCE_NONE ? d->dim : 0);
  g.kpad = 64;  // refined by lb_gns_create (32 or 64)

  const int64_t BN = e->BN;
  const size_t nc = (size_t)g.B * g.ncells;
  int rc = LB_OK;
  auto A = [&](int r) { if (!rc) rc = r; };
  A(lb_alloc(&e->win, (size_t)d->isl * d->dim * BN));
  A(lb_alloc(&e->ptype, (size_t)BN));
  A(lb_alloc(&e->ctrl, 1));
  A(lb_alloc(&e->blocks_done, 1));
  A(lb_alloc(&e->cell_of, (size_t)BN));
  A(lb_alloc(&e->cell_count, 2 * nc));  // cell_count | cell_fill contiguous (one memset)
  A(lb_alloc(&e->cell_start, nc + 1));
  A(lb_alloc(&e->cell_part, (size_t)BN));
  A(lb_alloc(&e->deg, (size_t)BN));
  A(lb_alloc(&e->nl_wg_sum, (size_t)BN / 8 + 2));
  A(lb_alloc(&e->row_ptr, (size_t)BN + 1));
  A(lb_alloc(&e->scan_part, (size_t)((BN > (int64_t)nc ? BN : (int64_t)nc) / 2048 + 2)));
  A(lb_alloc(&e->cpos, (size_t)d->dim * BN));
  e->cell_slots = BN;
  A(lb_alloc(&e->overflow, (size_t)g.B));
  A(lb_alloc(&e->nedges_b, (size_t)g.B));
  A(lb_alloc(&e->acc, (size_t)BN * 4));
  if (rc) { lb_engine_destroy(e); return rc; }
  e->cell_fill = e->cell_count + nc;
  if (hipHostMalloc((void**)&e->ctrl_host, sizeof(lb_ctrl)) != hipSuccess) {
    lb_engine_destroy(e);
    return lb_fail(LB_ERR_HIP, "hipHostMalloc failed");
  }
  if (hipHostMalloc((void**)&e->host_flag, sizeof(int32_t) * 4, hipHostMallocMapped) != hipSuccess ||
      hipHostGetDevicePointer((void**)&e->host_flag_dev, e->host_flag, 0) != hipSuccess) {
    lb_engine_destroy(e);
    return lb_fail(LB_ERR_HIP, "hipHostMalloc(mapped flag) failed");
  }
  e->host_flag[0] = -1;
  for (int i = 0; i < 4; ++i)
    if (hipEventCreateWithFlags(&e->step_ev[i], hipEventDisableTiming) != hipSuccess) {
      lb_engine_destroy(e);
      return lb_fail(LB_ERR_HIP, "hipEventCreate failed");
    }
  lb_ctrl c0;
  memset(&c0, 0, sizeof(c0));
  c0.overflow_step = -1;
  c0.math_step = LB_MATH_NO_STEP;
  c0.persist_step = LB_MATH_NO_STEP;
  c0.nl_epoch = 1;
  c0.ln_inv_d = 1.0f / LB_D;
  if (hipMemcpy(e->ctrl, &c0, sizeof(c0), hipMemcpyHostToDevice) != hipSuccess ||
      hipMemset(e->blocks_done, 0, sizeof(int32_t)) != hipSuccess ||
      hipMemset(e->ptype, 0, sizeof(int32_t) * BN) != hipSuccess ||
      hipMemset(e->nl_wg_sum, 0, sizeof(unsigned long long) * (BN / 8 + 2)) != hipSuccess ||
      hipMemset(e->row_ptr, 0, sizeof(int32_t) * (BN + 1)) != hipSuccess) {
    lb_engine_destroy(e);
    return lb_fail(LB_ERR_HIP, "engine init copies failed");
  }
  *out = e;
  return LB_OK;
}

extern "C" void lb_engine_destroy(lb_engine* e) {
  if (!e) return;
  lb_timers_collect(e);
  for (auto ev : e->epool) (void)hipEventDestroy(ev);
  void* bufs[] = {e->win, e->ptype, e->force, e->ctrl, e->cell_of, e->cell_count, e->cell_start,
                  e->cell_part, e->deg, e->nl_wg_sum, e->row_ptr, e->scan_part, e->cpos, e->tmp_send, e->tmp_feat, e->tmp_feat64,
                  e->senders, e->receivers, e->efeat, e->efeat64,
                  e->overflow, e->nedges_b, e->xnode, e->nlat, e->agg, e->psr, e->elat, e->msg,
                  e->acc, e->blocks_done};  // (e->part lives inside the e->agg allocation)
  for (void* b : bufs)
    if (b) (void)hipFree(b);
  if (e->ctrl_host) (void)hipHostFree(e->ctrl_host);
  if (e->host_flag) (void)hipHostFree(e->host_flag);
  for (int i = 0; i < 4; ++i)
    if (e->step_ev[i]) (void)hipEventDestroy(e->step_ev[i]);
  if (e->gstream) (void)hipStreamDestroy(e->gstream);
  delete e;
}

// Aggregated messages [BN][128] and the partial sums of the receivers a tile (chunk) boundary cuts [tiles + 2][2][128] in ONE
// allocation: the fused segment_sum of the edge kernels stores both through one buffer resource with branch-free
// raw-buffer stores (a lane with nothing to write has its offset pushed out of range).  Contents are transient (one layer).
int lb_alloc_aggpart(lb_engine* e) {
  if (e->agg) (void)hipFree(e->agg);
  e->agg = e->part = nullptr;
  const size_t n_agg = (size_t)e->BN * LB_D, n_part = (size_t)(e->e_alloc / 16 + 2) * 2 * LB_D;
  LB_TRY(lb_alloc(&e->agg, n_agg + n_part));
  e->part = e->agg + n_agg;
  e->aggpart_bytes = (int64_t)(n_agg + n_part) * (int64_t)sizeof(float);
  return LB_OK;
}

// Edge-sized buffers grow geometrically; contents are not preserved (rebuilt every step).
int lb_ensure_edges(lb_engine* e, int64_t need) {
  if (need <= e->e_alloc && e->senders) return LB_OK;
  LB_HIP(hipStreamSynchronize(e->stream));
  int64_t n = std::max<int64_t>(need + need / 8 + 1024, 4096);
  void* old[] = {e->senders, e->receivers, e->efeat, e->efeat64, e->elat, e->msg};
  for (void* b : old)
    if (b) (void)hipFree(b);
  e->senders = e->receivers = nullptr;
  e->efeat = nullptr;
  e->efeat64 = nullptr;
  e->elat = e->msg = nullptr;
  e->e_alloc = n;
  LB_TRY(lb_alloc_aggpart(e));  // agg | part (sized for the 16-row tiles of the new capacity)
  LB_TRY(lb_alloc(&e->senders, (size_t)n));
  LB_TRY(lb_alloc(&e->receivers, (size_t)n));
  // (kernels may gather through index rows past the current edge count before they know it: keep every row a valid id)
  LB_HIP(hipMemsetAsync(e->senders, 0, sizeof(int32_t) * (size_t)n, e->stream));
  LB_HIP(hipMemsetAsync(e->receivers, 0, sizeof(int32_t) * (size_t)n, e->stream));
  LB_TRY(lb_alloc(&e->efeat, (size_t)n * 8));
  LB_TRY(lb_alloc(&e->efeat64, (size_t)n * 4));
  LB_TRY(lb_alloc(&e->elat, (size_t)(n + 32) * LB_D));  // tile-blocked in the 16-row kernels: pad to a tile
  LB_TRY(lb_alloc(&e->msg, (size_t)(n + 32) * LB_D));  // also the second edge-latent buffer of the ping-pong
  e->e_alloc = n;
  return LB_OK;
}

extern "C" int lb_set_particle_type(lb_engine* e, const int32_t* ptype_dev) {
  if (!e || !ptype_dev) return lb_fail(LB_ERR_ARG, "null argument");
  LB_HIP(hipMemcpyAsync(e->ptype, ptype_dev, sizeof(int32_t) * e->BN, hipMemcpyDeviceToDevice,
                        e->stream));
  return LB_OK;
}

extern "C" int lb_set_force(lb_engine* e, const double* force_dev) {
  if (!e || !force_dev) return lb_fail(LB_ERR_ARG, "null argument");
  if (e->g.force_kind != LB_FORCE_BUFFER)
    return lb_fail(LB_ERR_STATE, "engine was not created with LB_FORCE_BUFFER");
  if (!e->force) LB_TRY(lb_alloc(&e->force, (size_t)e->BN * e->g.dim));
  LB_HIP(hipMemcpyAsync(e->force, force_dev, sizeof(double) * e->BN * e->g.dim,
                        hipMemcpyDeviceToDevice, e->stream));
  return LB_OK;
}

extern "C" int lb_load_window(lb_engine* e, const double* traj_dev, int32_t T, int32_t t0,
                              int32_t step) {
  if (!e || !traj_dev) return lb_fail(LB_ERR_ARG, "null argument");
  if (t0 < 0 || t0 + e->g.isl > T) return lb_fail(LB_ERR_ARG, "window [%d,%d) outside T=%d", t0, t0 + e->g.isl, T);
  return lbk_load_window(e, traj_dev, T, t0, step);
}

extern "C" int lb_read_window(lb_engine* e, double* out) {
  if (!e || !out) return lb_fail(LB_ERR_ARG, "null argument");
  return lbk_read_window(e, out);
}

// --------------------------------------------------------------------------- neighbor list
static int lb_check_density(lb_engine* e) {
  if (e->ctrl_host->persist_error == 2) {
    e->nl_one_off = true;  // k_nl_small gave up waiting for a predecessor workgroup: back to the multi-launch build
    (void)hipMemsetAsync(&e->ctrl->persist_error, 0, sizeof(int32_t), e->stream);
    return lb_fail(LB_ERR_STATE, "single-launch neighbor build: a predecessor workgroup's edge count never arrived; "
                                 "results of this call are invalid, the engine now uses the multi-launch build");
  }
  if (e->ctrl_host->density_error)
    return lb_fail(LB_ERR_DENSITY,
                   "neighbor search: %s (limits of the dense fall-back: %d neighbors per particle)",
                   e->ctrl_host->density_error == 1 ? "3^dim-cell stencil too populated"
                                                    : "a particle has too many neighbors",
                   LB_MAX_ROW_DENSE);
  return LB_OK;
}

extern "C" int lb_nl_allocate(lb_engine* e, int32_t* cell_capacity_out, int32_t* e_cap_out,
                              int32_t* occupancy_out) {
  if (!e) return lb_fail(LB_ERR_ARG, "null engine");
  // clear the poison + thaw the capacities: this is an allocation, not an update
  lb_ctrl* h = e->ctrl_host;
  LB_HIP(hipMemcpyAsync(h, e->ctrl, sizeof(lb_ctrl), hipMemcpyDeviceToHost, e->stream));
  LB_HIP(hipStreamSynchronize(e->stream));
  h->overflow_step = -1;
  h->density_error = 0;
  e->host_flag[0] = -1;
  LB_HIP(hipMemcpyAsync(e->ctrl, h, sizeof(lb_ctrl), hipMemcpyHostToDevice, e->stream));
  e->cell_capacity = 0;
  e->e_cap = 0;
  LB_TRY(lbk_nl_build(e, true));
  std::vector<int32_t> occ(e->g.B);
  LB_HIP(hipMemcpyAsync(h, e->ctrl, sizeof(lb_ctrl), hipMemcpyDeviceToHost, e->stream));
  LB_HIP(hipMemcpyAsync(occ.data(), e->nedges_b, sizeof(int32_t) * e->g.B, hipMemcpyDeviceToHost,
                        e->stream));
  LB_HIP(hipStreamSynchronize(e->stream));
  LB_TRY(lb_check_density(e));
  const double mult = e->desc.capacity_multiplier > 0 ? e->desc.capacity_multiplier : 1.25;
  // jax-md: cell_capacity = int(max_cell_occupancy * multiplier)
  e->cell_capacity = e->g.use_cell_list ? (int)(h->max_cell_occ * mult) : 0;
  int32_t occ_max = 0;
  for (int b = 0; b < e->g.B; ++b) occ_max = std::max(occ_max, occ[b]);
  // jax-md: max_occupancy = int(occupancy * multiplier), clamped to the candidate count and N^2
  int64_t ecap = (int64_t)(occ_max * mult);
  const int64_t N = e->g.N;
  int64_t cand = e->g.use_cell_list ? N * e->g.nstencil * std::max(e->cell_capacity, 1) : N * N;
  ecap = std::min(ecap, cand);
  ecap = std::min(ecap, N * N);
  if (ecap * e->g.B > (int64_t)1 << 30) return lb_fail(LB_ERR_ARG, "B*E_cap exceeds int32 range");
  e->e_cap = (int32_t)std::max<int64_t>(ecap, 1);
  LB_TRY(lb_ensure_edges(e, (int64_t)e->e_cap * e->g.B));
  // per-node slots of the single-sweep update path: twice the current max degree
  {
    int32_t want = std::max(16, ((2 * h->max_deg + 7) / 8) * 8);
    want = std::min<int32_t>(want, e->nl_dense ? LB_MAX_ROW_DENSE : LB_MAX_ROW);
    if (e->nl_dense) e->row_cap = std::max(e->row_cap, std::min<int32_t>((want + 63) / 64 * 64, LB_MAX_ROW_DENSE));
    if (want > e->maxd || !e->tmp_send) {
      LB_HIP(hipStreamSynchronize(e->stream));
      for (void* b : {(void*)e->tmp_send, (void*)e->tmp_feat, (void*)e->tmp_feat64})
        if (b) (void)hipFree(b);
      e->tmp_send = nullptr;
      e->tmp_feat = nullptr;
      e->tmp_feat64 = nullptr;
      e->maxd = want;
      LB_TRY(lb_alloc(&e->tmp_send, (size_t)e->BN * want));
      LB_TRY(lb_alloc(&e->tmp_feat, (size_t)e->BN * want * 4));
      LB_TRY(lb_alloc(&e->tmp_feat64, (size_t)e->BN * want * 4));
    }
  }
  if (cell_capacity_out) *cell_capacity_out = e->cell_capacity;
  if (e_cap_out) *e_cap_out = e->e_cap;
  if (occupancy_out) memcpy(occupancy_out, occ.data(), sizeof(int32_t) * e->g.B);
  return LB_OK;
}

extern "C" int lb_nl_set_capacity(lb_engine* e, int32_t cell_capacity, int32_t e_cap) {
  if (!e || e_cap < 1) return lb_fail(LB_ERR_ARG, "bad capacity");
  e->cell_capacity = cell_capacity;
  e->e_cap = e_cap;
  return lb_ensure_edges(e, (int64_t)e_cap * e->g.B);
}

extern "C" int lb_nl_update(lb_engine* e) {
  if (!e) return lb_fail(LB_ERR_ARG, "null engine");
  if (e->e_cap <= 0) return lb_fail(LB_ERR_STATE, "lb_nl_update before lb_nl_allocate");
  return lbk_nl_build(e, true);
}

extern "C" int lb_nl_read_flags(lb_engine* e, int32_t* overflow_out_dev) {
  if (!e || !overflow_out_dev) return lb_fail(LB_ERR_ARG, "null argument");
  LB_HIP(hipMemcpyAsync(overflow_out_dev, e->overflow, sizeof(int32_t) * e->g.B,
                        hipMemcpyDeviceToDevice, e->stream));
  return LB_OK;
}

extern "C" int lb_nl_read_idx(lb_engine* e, int32_t* idx_out_dev, int32_t* n_edges_out_dev) {
  if (!e || !idx_out_dev) return lb_fail(LB_ERR_ARG, "null argument");
  if (e->e_cap <= 0) return lb_fail(LB_ERR_STATE, "no neighbor list allocated");
  return lbk_nl_export(e, idx_out_dev, n_edges_out_dev);
}

extern "C" int lb_node_features(lb_engine* e, double* vel_hist, double* vel_mag, double* bound,
                                double* force) {
  if (!e || !vel_hist) return lb_fail(LB_ERR_ARG, "null argument");
  if (e->g.force_kind == LB_FORCE_BUFFER && !e->force)
    return lb_fail(LB_ERR_STATE, "LB_FORCE_BUFFER engine: call lb_set_force first");
  return lbk_node_features(e, nullptr, nullptr, 0, 0, vel_hist, e->g.has_vel_mag ? vel_mag : nullptr,
                           e->g.has_bound ? bound : nullptr,
                           e->g.force_kind != LB_FORCE_NONE ? force : nullptr);
}

extern "C" int lb_edge_features(lb_engine* e, double* rel_disp, double* rel_dist) {
  if (!e || !rel_disp || !rel_dist) return lb_fail(LB_ERR_ARG, "null argument");
  if (e->e_cap <= 0) return lb_fail(LB_ERR_STATE, "no neighbor list allocated");
  return lbk_edge_features_export(e, rel_disp, rel_dist);
}

extern "C" int lb_stats(lb_engine* e, int64_t* n_edges_total, int32_t* e_cap, int32_t* cell_capacity) {
  if (!e) return lb_fail(LB_ERR_ARG, "null engine");
  LB_HIP(hipMemcpyAsync(e->ctrl_host, e->ctrl, sizeof(lb_ctrl), hipMemcpyDeviceToHost, e->stream));
  LB_HIP(hipStreamSynchronize(e->stream));
  if (n_edges_total) *n_edges_total = e->ctrl_host->n_edges_unclamped;
  if (e_cap) *e_cap = e->e_cap;
  if (cell_capacity) *cell_capacity = e->cell_capacity;
  return LB_OK;
}

extern "C" int lb_edge_accounting(lb_engine* e, int64_t* sum_edges, int64_t* n_builds, int64_t* first_edges,
                                  int64_t* last_edges, int32_t reset) {
  if (!e) return lb_fail(LB_ERR_ARG, "null engine");
  LB_HIP(hipMemcpyAsync(e->ctrl_host, e->ctrl, sizeof(lb_ctrl), hipMemcpyDeviceToHost, e->stream));
  LB_HIP(hipStreamSynchronize(e->stream));
  if (sum_edges) *sum_edges = e->ctrl_host->acct_sum;
  if (n_builds) *n_builds = e->ctrl_host->acct_builds;
  if (first_edges) *first_edges = e->ctrl_host->acct_first;
  if (last_edges) *last_edges = e->ctrl_host->n_edges_unclamped;
  if (reset) {
    // acct_builds, acct_first, (padding,) acct_sum: everything from acct_builds to the end of acct_sum (the
    // 8-aligned acct_sum sits behind 4 bytes of padding, a 16-byte memset left its high half alive)
    static_assert(offsetof(lb_ctrl, acct_first) == offsetof(lb_ctrl, acct_builds) + sizeof(int32_t) &&
                      offsetof(lb_ctrl, acct_sum) > offsetof(lb_ctrl, acct_first),
                  "lb_ctrl accounting fields must be contiguous: acct_builds, acct_first, acct_sum");
    LB_HIP(hipMemsetAsync(&e->ctrl->acct_builds, 0,
                          offsetof(lb_ctrl, acct_sum) + sizeof(int64_t) - offsetof(lb_ctrl, acct_builds), e->stream));
    LB_HIP(hipStreamSynchronize(e->stream));
  }
  return LB_OK;
}

extern "C" int lb_segment_sum(lb_engine* e, const float* msg_dev, float* out_dev, int32_t D) {
  if (!e || !msg_dev || !out_dev) return lb_fail(LB_ERR_ARG, "null argument");
  lb_tic(e, LB_T_AGGREGATE);
  int rc = lbk_segment_sum(e, msg_dev, out_dev, D);
  lb_toc(e);
  return rc;
}

// ------------------------------------------------------------------------------------- GNS
extern "C" int lb_gns_create(lb_engine* e, const lb_gns_desc* d, const float* w, int64_t n_floats,
                             lb_gns** out) {
  if (!e || !d || !w || !out) return lb_fail(LB_ERR_ARG, "null argument");
  if (d->latent_size < 16 || d->latent_size > LB_D || d->latent_size % 16)
    return lb_fail(LB_ERR_UNSUPPORTED, "latent_size %d not built (multiples of 16 up to 128)", d->latent_size);
  if (d->out_dim != e->g.dim) return lb_fail(LB_ERR_ARG, "out_dim %d != case dim %d", d->out_dim, e->g.dim);
  if (d->node_in != e->g.node_in) return lb_fail(LB_ERR_ARG, "node_in %d != case feature width %d", d->node_in, e->g.node_in);
  if (d->edge_in != e->g.dim + 1) return lb_fail(LB_ERR_ARG, "edge_in %d != dim+1", d->edge_in);
  if (d->num_mp_steps < 0 || d->num_mp_steps > 64) return lb_fail(LB_ERR_ARG, "bad num_mp_steps");
  // any MLP depth other than the published two Linears runs on the one-Linear-per-launch kernels
  if (d->blocks_per_step != 2) return lb_gns_create_generic(e, d, w, n_floats, out);
  const int D = LB_D, L = d->num_mp_steps;
  const bool has_emb = d->num_particle_types > 1;
  const int emb = has_emb ? d->embedding_size : 0;
  const int nin = d->node_in + emb;
  if (nin > 128) return lb_fail(LB_ERR_UNSUPPORTED, "node input width %d > 128 not built", nin);
  const int kpad = (nin + 31) / 32 * 32;  // 32 .. 128

  // expected blob length
  auto mlp_len = [&](int in, int outw, bool ln) -> int64_t {
    return (int64_t)in * D + D + (int64_t)D * outw + outw + (ln ? 2 * outw : 0);
  };
  int64_t expect = (has_emb ? (int64_t)d->num_particle_types * emb : 0) + mlp_len(nin, D, true) +
                   mlp_len(d->edge_in, D, true) +
                   (int64_t)L * (mlp_len(3 * D, D, true) + mlp_len(2 * D, D, true)) +
                   mlp_len(D, d->out_dim, false);
  if (d->latent_size == D && expect != n_floats)
    return lb_fail(LB_ERR_ARG, "weight blob has %lld floats, expected %lld", (long long)n_floats, (long long)expect);

  // A latent narrower than the 128-wide tiles (GNS-5-64, docs/pages/baselines.rst:54) runs on the same
  // kernels: every Linear is zero-padded to 128 columns / its input blocks to 128-row strides, LayerNorm
  // scale / offset are padded with zeros (the padded features stay exactly 0 through every layer) and the
  // kernels divide by the true width (lb_ctrl::ln_inv_d / ln_pad).
  std::vector<float> widened;
  const int dl = d->latent_size;
  if (dl != D) {
    auto mlp_len_d = [&](int in, int outw, bool ln) -> int64_t {
      return (int64_t)in * dl + dl + (int64_t)dl * outw + outw + (ln ? 2 * outw : 0);
    };
    const int64_t expect_d = (has_emb ? (int64_t)d->num_particle_types * emb : 0) + mlp_len_d(nin, dl, true) +
                             mlp_len_d(d->edge_in, dl, true) +
                             (int64_t)L * (mlp_len_d(3 * dl, dl, true) + mlp_len_d(2 * dl, dl, true)) +
                             mlp_len_d(dl, d->out_dim, false);
    if (expect_d != n_floats)
      return lb_fail(LB_ERR_ARG, "weight blob has %lld floats, expected %lld", (long long)n_floats, (long long)expect_d);
    const float* q = w;
    auto copy = [&](size_t n) {
      widened.insert(widened.end(), q, q + n);
      q += n;
    };
    // one Linear (in_blocks x blk_in rows, out columns) -> (in_blocks x blk_pad rows, out_pad columns)
    auto linear = [&](int in_blocks, int blk_in, int blk_pad, int out, int out_pad) {
      const size_t base = widened.size();
      widened.resize(base + (size_t)in_blocks * blk_pad * out_pad, 0.f);
      for (int b = 0; b < in_blocks; ++b)
        for (int r = 0; r < blk_in; ++r)
          for (int c = 0; c < out; ++c)
            widened[base + ((size_t)b * blk_pad + r) * out_pad + c] = q[((size_t)b * blk_in + r) * out + c];
      q += (size_t)in_blocks * blk_in * out;
    };
    auto vec = [&](int n, int n_pad) {
      const size_t base = widened.size();
      widened.resize(base + n_pad, 0.f);
      for (int i = 0; i < n; ++i) widened[base + i] = q[i];
      q += n;
    };
    auto mlp = [&](int in_blocks, int blk_in, int blk_pad, int outw, int outw_pad, bool ln) {
      linear(in_blocks, blk_in, blk_pad, dl, D);   // w0
      vec(dl, D);                                  // b0
      linear(1, dl, D, outw, outw_pad);            // w1
      vec(outw, outw_pad);                         // b1
      if (ln) {
        vec(outw, outw_pad);
        vec(outw, outw_pad);
      }
    };
    if (has_emb) copy((size_t)d->num_particle_types * emb);
    mlp(1, nin, nin, dl, D, true);
    mlp(1, d->edge_in, d->edge_in, dl, D, true);
    for (int k = 0; k < L; ++k) {
      mlp(3, dl, D, dl, D, true);
      mlp(2, dl, D, dl, D, true);
    }
    mlp(1, dl, D, d->out_dim, d->out_dim, false);
    if (q - w != n_floats) return lb_fail(LB_ERR_ARG, "internal: widening walk mismatch");
    w = widened.data();
    n_floats = (int64_t)widened.size();
  }

  std::vector<float> host;
  auto put = [&](const float* src, size_t n) -> size_t {
    size_t off = host.size();
    off = (off + 63) & ~(size_t)63;  // 256-byte alignment of every block
    host.resize(off + n, 0.f);
    if (src) memcpy(host.data() + off, src, n * sizeof(float));
    return off;
  };
  auto put_packed16 = [&](const float* src, int K, int M, int Kp) -> size_t {
    std::vector<float> tmp((size_t)Kp * 128);
    lb_pack_weight16(src, K, M, Kp, tmp.data());
    return put(tmp.data(), tmp.size());
  };
  auto put_packed16h = [&](const float* src, int K, int M, int Kp, int Mp = 128) -> size_t {
    std::vector<float> tmp((size_t)Kp * Mp);
    lb_pack_weight16h(src, K, M, Kp, tmp.data(), Mp);
    return put(tmp.data(), tmp.size());
  };
  auto put_packed = [&](const float* src, int K, int M, int Kp, int Mp) -> size_t {
    std::vector<float> tmp((size_t)Kp * Mp);
    lb_pack_weight(src, K, M, Kp, Mp, tmp.data());
    return put(tmp.data(), tmp.size());
  };
  struct Off { size_t w0, b0, w1, b1, lns, lno; bool ln; };
  const float* p = w;
  size_t off_embed = 0;
  if (has_emb) {
    off_embed = put(p, (size_t)d->num_particle_types * emb);
    p += (size_t)d->num_particle_types * emb;
  }
  // f16x2 carries a weight as fp16 hi + fp16 lo with an ABSOLUTE floor of 2^-25 on the pair: a matrix whose
  // entries are uniformly small (rms < 2^-7) would lose the 1e-5 class - noted here, acted on below
  double w_rms_min = 1e30;
  auto note_rms = [&](const float* m, size_t n) {
    double s2 = 0;
    size_t nz = 0;
    for (size_t i = 0; i < n; ++i) {
      s2 += (double)m[i] * m[i];
      nz += m[i] != 0.f;
    }
    if (nz) w_rms_min = std::min(w_rms_min, std::sqrt(s2 / (double)nz));
  };
  // generic 2-layer MLP reader; k0pad = padded K of layer 0; outp = padded out width
  auto read_mlp = [&](int in, int k0pad, int outw, int outp, bool ln) -> Off {
    Off o{};
    note_rms(p, (size_t)in * D);
    o.w0 = put_packed(p, in, D, k0pad, D); p += (size_t)in * D;
    o.b0 = put(p, D); p += D;
    if (ln) note_rms(p, (size_t)D * outw);  // (the decoder's output Linear is rescaled instead, see below)
    o.w1 = put_packed(p, D, outw, D, outp); p += (size_t)D * outw;
    {
      std::vector<float> b(outp, 0.f);
      memcpy(b.data(), p, sizeof(float) * outw);
      o.b1 = put(b.data(), outp);
      p += outw;
    }
    o.ln = ln;
    if (ln) {
      o.lns = put(p, outw); p += outw;
      o.lno = put(p, outw); p += outw;
    }
    return o;
  };
  const size_t o_en_w0_h = put_packed16h(p, nin, D, kpad);
  const size_t o_en_w1_h = put_packed16h(p + (size_t)nin * D + D, D, D, D);
  std::vector<size_t> o_pn_w0_h(L), o_pn_w1_h(L), o_pw_h(L), o_pw_h2(L);
  Off o_enc_node = read_mlp(nin, kpad, D, D, true);
  const float* p_enc_edge = p;
  Off o_enc_edge = read_mlp(d->edge_in, 8, D, D, true);
  const size_t o_ee_w0_16 = put_packed16(p_enc_edge, d->edge_in, D, 16);
  const size_t o_ee_w1_16 = put_packed16(p_enc_edge + (size_t)d->edge_in * D + D, D, D, D);
  const size_t o_ee_w0_16h = put_packed16h(p_enc_edge, d->edge_in, D, 32);
  const size_t o_ee_w1_16h = put_packed16h(p_enc_edge + (size_t)d->edge_in * D + D, D, D, D);
  std::vector<size_t> o_pe_w0_16(L), o_pe_w1_16(L), o_pe_w0_16h(L), o_pe_w1_16h(L);
  std::vector<Off> o_pe(L), o_pn(L);
  std::vector<size_t> o_pw(L), o_pb(L);
  for (int k = 0; k < L; ++k) {
    // edge MLP: w0 is (3D, D) over [sender | receiver | edge] (gns.py:97-100)
    const float* w0 = p;
    const float* b0 = p + (size_t)3 * D * D;
    // projection for the node kernel: (D, 2D) = [Ws | Wr], bias [0 | b0]
    {
      std::vector<float> wsr((size_t)D * 2 * D);
      for (int kk = 0; kk < D; ++kk)
        for (int m = 0; m < D; ++m) {
          wsr[(size_t)kk * 2 * D + m] = w0[(size_t)kk * D + m];
          wsr[(size_t)kk * 2 * D + D + m] = w0[(size_t)(D + kk) * D + m];
        }
      o_pw[k] = put_packed(wsr.data(), D, 2 * D, D, 2 * D);
      o_pw_h[k] = put_packed16h(wsr.data(), D, 2 * D, D, 2 * D);
      {  // the same projection as two 128-wide halves [Ws | Wr]: uniform 32 KiB chunks for lb_node16s.hip
        std::vector<float> two((size_t)2 * D * D);
        lb_pack_weight16h(w0, D, D, D, two.data(), D);
        lb_pack_weight16h(w0 + (size_t)D * D, D, D, D, two.data() + (size_t)D * D, D);
        o_pw_h2[k] = put(two.data(), two.size());
      }
      std::vector<float> bb(2 * D, 0.f);
      memcpy(bb.data() + D, b0, sizeof(float) * D);
      o_pb[k] = put(bb.data(), 2 * D);
    }
    Off o{};
    o.w0 = put_packed(w0 + (size_t)2 * D * D, D, D, D, D);  // edge rows only
    o_pe_w0_16[k] = put_packed16(w0 + (size_t)2 * D * D, D, D, D);
    o_pe_w0_16h[k] = put_packed16h(w0 + (size_t)2 * D * D, D, D, D);
    o.b0 = put(b0, D);
    p += (size_t)3 * D * D + D;
    o_pe_w1_16[k] = put_packed16(p, D, D, D);
    o_pe_w1_16h[k] = put_packed16h(p, D, D, D);
    o.w1 = put_packed(p, D, D, D, D); p += (size_t)D * D;
    o.b1 = put(p, D); p += D;
    o.ln = true;
    o.lns = put(p, D); p += D;
    o.lno = put(p, D); p += D;
    o_pe[k] = o;
    o_pn_w0_h[k] = put_packed16h(p, 2 * D, D, 2 * D);
    o_pn_w1_h[k] = put_packed16h(p + (size_t)2 * D * D + D, D, D, D);
    o_pn[k] = read_mlp(2 * D, 2 * D, D, D, true);
  }
  const float* p_dec = p;
  Off o_dec = read_mlp(D, D, d->out_dim, 32, false);
  const size_t o_dec_w0_h = put_packed16h(p_dec, D, D, D);
  const size_t o_dec_w0_f = put_packed16(p_dec, D, D, D);
  // decoder head: the f16x2 copy is packed times 2^s (max |w| -> [0.25, 0.5)) and the kernel multiplies the result
  // by 2^-s: exact, and independent of the output normalisation a checkpoint was trained with
  float dec_unscale = 1.f;
  size_t o_dec_w1_h;
  {
    const float* w1d = p_dec + (size_t)D * D + D;
    const size_t n = (size_t)D * d->out_dim;
    float mx = 0.f;
    for (size_t i = 0; i < n; ++i) mx = std::max(mx, std::fabs(w1d[i]));
    int sh = 0;
    if (mx > 0.f && std::isfinite(mx)) sh = std::max(-60, std::min(60, (int)std::floor(std::log2(0.5 / (double)mx))));
    std::vector<float> scaled(w1d, w1d + n);
    for (float& x : scaled) x = std::ldexp(x, sh);
    dec_unscale = std::ldexp(1.f, -sh);
    o_dec_w1_h = put_packed16h(scaled.data(), D, d->out_dim, D, 16);
  }
  const size_t o_dec_w1_f = put_packed16(p_dec + (size_t)D * D + D, D, d->out_dim, D);
  if (p - w != n_floats) return lb_fail(LB_ERR_ARG, "internal: blob walk mismatch");
  // M-split images (lb_msplit.hip): one block per MLP = [W0 | W1 | projection of the next edge MLP]
  auto put_ms = [&](std::initializer_list<std::tuple<const float*, int, int, int, int, bool>> mats) -> size_t {
    std::vector<float> img;
    for (const auto& m : mats) {
      const float* src; int K, M, nkb, npw; bool perm;
      std::tie(src, K, M, nkb, npw, perm) = m;
      const size_t base = img.size();
      img.resize(base + (size_t)nkb * npw * 4096);
      lb_pack_ms(src, K, M, nkb, npw, perm, img.data() + base);
    }
    return put(img.data(), img.size());
  };
  // walk the blob once more for the source pointers (same order as above)
  size_t o_ms_enc_node, o_ms_enc_edge;
  std::vector<size_t> o_ms_pe(L), o_ms_pn(L);
  {
    const float* q = w + (has_emb ? (size_t)d->num_particle_types * emb : 0);
    auto wsr_of = [&](const float* w0e, std::vector<float>& wsr) {  // [Ws | Wr] of an edge MLP's first Linear
      wsr.assign((size_t)D * 2 * D, 0.f);
      for (int kk = 0; kk < D; ++kk)
        for (int m = 0; m < D; ++m) {
          wsr[(size_t)kk * 2 * D + m] = w0e[(size_t)kk * D + m];
          wsr[(size_t)kk * 2 * D + D + m] = w0e[(size_t)(D + kk) * D + m];
        }
    };
    const float* en_w0 = q;
    const float* en_w1 = q + (size_t)nin * D + D;
    q += (size_t)nin * D + D + (size_t)D * D + D + 2 * D;
    const float* ee_w0 = q;
    const float* ee_w1 = q + (size_t)d->edge_in * D + D;
    q += (size_t)d->edge_in * D + D + (size_t)D * D + D + 2 * D;
    std::vector<const float*> pe_w0(L), pe_w1(L), pn_w0(L), pn_w1(L);
    for (int k = 0; k < L; ++k) {
      pe_w0[k] = q;
      pe_w1[k] = q + (size_t)3 * D * D + D;
      q += (size_t)3 * D * D + D + (size_t)D * D + D + 2 * D;
      pn_w0[k] = q;
      pn_w1[k] = q + (size_t)2 * D * D + D;
      q += (size_t)2 * D * D + D + (size_t)D * D + D + 2 * D;
    }
    std::vector<float> wsr;
    if (L > 0) {
      wsr_of(pe_w0[0], wsr);
      o_ms_enc_node = put_ms({{en_w0, nin, D, kpad / 32, 1, true}, {en_w1, D, D, 4, 1, true}, {wsr.data(), D, 2 * D, 4, 2, true}});
    } else {
      o_ms_enc_node = put_ms({{en_w0, nin, D, kpad / 32, 1, true}, {en_w1, D, D, 4, 1, true}});
    }
    o_ms_enc_edge = put_ms({{ee_w0, d->edge_in, D, 1, 1, false}, {ee_w1, D, D, 4, 1, true}});
    for (int k = 0; k < L; ++k) {
      o_ms_pe[k] = put_ms({{pe_w0[k] + (size_t)2 * D * D, D, D, 4, 1, true}, {pe_w1[k], D, D, 4, 1, true}});
      if (k + 1 < L) {
        wsr_of(pe_w0[k + 1], wsr);
        o_ms_pn[k] = put_ms({{pn_w0[k], 2 * D, D, 8, 1, true}, {pn_w1[k], D, D, 4, 1, true}, {wsr.data(), D, 2 * D, 4, 2, true}});
      } else {
        // last layer: [W0 | W1 | decoder W0 | decoder W1 (out_dim block, scaled like dec_w1_h)] - k_node_ms<DEC>
        std::vector<float> img((size_t)8 * 4096 + 4 * 4096 + 4 * 4096 + 2048, 0.f), tmp((size_t)4 * 4096);
        lb_pack_ms(pn_w0[k], 2 * D, D, 8, 1, true, img.data());
        lb_pack_ms(pn_w1[k], D, D, 4, 1, true, img.data() + (size_t)8 * 4096);
        lb_pack_ms(p_dec, D, D, 4, 1, true, img.data() + (size_t)12 * 4096);
        {
          const float* w1d = p_dec + (size_t)D * D + D;
          std::vector<float> scaled(w1d, w1d + (size_t)D * d->out_dim);
          for (float& x : scaled) x = x / dec_unscale;  // dec_unscale is a power of two: exact
          lb_pack_ms(scaled.data(), D, d->out_dim, 4, 1, true, tmp.data());
          memcpy(img.data() + (size_t)16 * 4096, tmp.data(), sizeof(float) * 2048);  // block 0 = outputs 0 .. 15
        }
        o_ms_pn[k] = put(img.data(), img.size());
      }
    }
  }

  lb_gns* g = new lb_gns();
  g->desc = *d;
  g->eng = e;
  g->tap = nullptr;
  g->kq_node = kpad / 8;
  if (hipMalloc((void**)&g->blob, host.size() * sizeof(float)) != hipSuccess) {
    delete g;
    return lb_fail(LB_ERR_HIP, "hipMalloc(weights) failed");
  }
  if (hipMemcpy(g->blob, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
    (void)hipFree(g->blob);
    delete g;
    return lb_fail(LB_ERR_HIP, "weight upload failed");
  }
  auto mk = [&](const Off& o) {
    lb_mlp_w m{};
    m.w0 = g->blob + o.w0;
    m.b0 = g->blob + o.b0;
    m.w1 = g->blob + o.w1;
    m.b1 = g->blob + o.b1;
    m.ln_s = o.ln ? g->blob + o.lns : nullptr;
    m.ln_o = o.ln ? g->blob + o.lno : nullptr;
    return m;
  };
  g->embed = has_emb ? g->blob + off_embed : nullptr;
  g->enc_node = mk(o_enc_node);
  g->enc_edge = mk(o_enc_edge);
  g->dec = mk(o_dec);
  for (int k = 0; k < L; ++k) {
    g->proc_edge.push_back(mk(o_pe[k]));
    g->proc_node.push_back(mk(o_pn[k]));
    g->proj_w.push_back(g->blob + o_pw[k]);
    g->proj_b.push_back(g->blob + o_pb[k]);
    g->proc_edge_w0_16.push_back(g->blob + o_pe_w0_16[k]);
    g->proc_edge_w1_16.push_back(g->blob + o_pe_w1_16[k]);
    g->proc_node_w0_h.push_back(g->blob + o_pn_w0_h[k]);
    g->proc_node_w1_h.push_back(g->blob + o_pn_w1_h[k]);
    g->proj_w_h.push_back(g->blob + o_pw_h[k]);
    g->proj_w_h2.push_back(g->blob + o_pw_h2[k]);
    g->proc_edge_w0_16h.push_back(g->blob + o_pe_w0_16h[k]);
    g->proc_edge_w1_16h.push_back(g->blob + o_pe_w1_16h[k]);
  }
  g->ms_enc_node = g->blob + o_ms_enc_node;
  g->ms_enc_edge = g->blob + o_ms_enc_edge;
  for (int k = 0; k < L; ++k) {
    g->ms_proc_edge.push_back(g->blob + o_ms_pe[k]);
    g->ms_proc_node.push_back(g->blob + o_ms_pn[k]);
  }
  g->dec_unscale = dec_unscale;
  g->dec_w0_h = g->blob + o_dec_w0_h;
  g->dec_w0_f = g->blob + o_dec_w0_f;
  g->dec_w1_h = g->blob + o_dec_w1_h;
  g->dec_w1_f = g->blob + o_dec_w1_f;
  g->enc_edge_w0_16 = g->blob + o_ee_w0_16;
  g->enc_edge_w1_16 = g->blob + o_ee_w1_16;
  g->enc_node_w0_h = g->blob + o_en_w0_h;
  g->enc_node_w1_h = g->blob + o_en_w1_h;
  g->enc_edge_w0_16h = g->blob + o_ee_w0_16h;
  g->enc_edge_w1_16h = g->blob + o_ee_w1_16h;
  // LayerNorm width of this model (read by every network kernel through the control block: lb_gns_bind)
  g->lnc[0] = 1.0f / (float)dl;
  g->lnc[1] = (float)(D - dl);
  if (w_rms_min < 0.0078125 && e->f16x2 && e->math_auto) {
    fprintf(stderr, "[lbhip] a weight matrix has rms %.3g < 2^-7: its fp16 hi/lo split would fall short of the 1e-5 class - "
                    "this engine uses exact-fp32 MFMA arithmetic\n", w_rms_min);
    e->f16x2 = 0;
  }
  int rc = lb_ensure_node_scratch(e);
  if (!rc) rc = lb_gns_bind(e, g);
  if (rc) {
    lb_gns_destroy(g);
    return rc;
  }
  *out = g;
  return LB_OK;
}

int lb_ensure_node_scratch(lb_engine* e) {
  const int64_t BN = e->BN;
  if (!e->xnode) LB_TRY(lb_alloc(&e->xnode, (size_t)BN * LB_D));  // widest node input row
  if (!e->nlat) LB_TRY(lb_alloc(&e->nlat, (size_t)BN * LB_D));
  if (!e->agg) LB_TRY(lb_alloc_aggpart(e));
  if (!e->psr) LB_TRY(lb_alloc(&e->psr, (size_t)BN * 2 * LB_D));
  return LB_OK;
}

// Per-model constants that live in engine-wide state (the LayerNorm width in the control block, the node feature
// row stride in the geometry): re-applied whenever another model of the same engine runs.
int lb_gns_bind(lb_engine* e, lb_gns* g) {
  if (e->bound_model == g) return LB_OK;
  LB_HIP(hipMemcpyAsync(&e->ctrl->ln_inv_d, g->lnc, sizeof(g->lnc), hipMemcpyHostToDevice, e->stream));
  e->g.kpad = g->kq_node * 8;
  e->bound_model = g;
  return LB_OK;
}

extern "C" void lb_gns_destroy(lb_gns* g) {
  if (!g) return;
  if (g->eng && g->eng->bound_model == g) g->eng->bound_model = nullptr;
  if (g->blob) (void)hipFree(g->blob);
  for (float* b : g->gen_hn)
    if (b) (void)hipFree(b);
  if (g->gen_he) (void)hipFree(g->gen_he);
  delete g;
}

extern "C" int lb_set_fused_aggregation(lb_engine* e, int32_t on) {
  if (!e) return lb_fail(LB_ERR_ARG, "null engine");
  e->fused_agg = on ? 1 : 0;
  return LB_OK;
}

extern "C" int lb_gns_set_tap(lb_gns* g, float* tap) {
  if (!g) return lb_fail(LB_ERR_ARG, "null model");
  g->tap = tap;
  return LB_OK;
}

__global__ void k_acc_export(int64_t BN, int dim, const float* __restrict__ acc4,
                             float* __restrict__ out) {
  int64_t gi = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gi >= BN) return;
  for (int d = 0; d < dim; ++d) out[gi * dim + d] = acc4[gi * 4 + d];
}

// f16x2 range guard, host side: read (and clear) the flags the kernels raised since the last check.
// *switched = 1 when the engine was in guarded f16x2 mode and has just been put on exact fp32 - the caller then
// repeats its work and sets f16x2 back afterwards (not sticky).  Called with the engine already on fp32 it only
// clears the flags (the fp32 decoder raises LB_MATH_NONFINITE too) and reports them through flags_out.
static int lb_math_check(lb_engine* e, int* switched, int* flagged_step = nullptr, int* flags_out = nullptr) {
  *switched = 0;
  if (flags_out) *flags_out = 0;
  LB_HIP(hipMemcpyAsync(e->ctrl_host, e->ctrl, sizeof(lb_ctrl), hipMemcpyDeviceToHost, e->stream));
  LB_HIP(hipStreamSynchronize(e->stream));
  const int flags = e->ctrl_host->math_flags;
  if (!flags) return LB_OK;
  if (flags_out) *flags_out = flags;
  if (flagged_step) *flagged_step = e->ctrl_host->math_step;
  {
    const int32_t reset[1] = {LB_MATH_NO_STEP};
    LB_HIP(hipMemsetAsync(&e->ctrl->math_flags, 0, sizeof(int32_t), e->stream));
    LB_HIP(hipMemcpyAsync(&e->ctrl->math_step, reset, sizeof(reset), hipMemcpyHostToDevice, e->stream));
    LB_HIP(hipStreamSynchronize(e->stream));
  }
  if (e->f16x2 && e->math_auto) {
    fprintf(stderr,
            "[lbhip] f16x2 range guard raised (%s%s%s): the flagged step is redone in exact-fp32 MFMA arithmetic\n",
            flags & LB_MATH_LARGE ? "operand >= 2^15 " : "", flags & LB_MATH_TINY ? "operand row / tile below the fp16 split's range " : "",
            flags & LB_MATH_NONFINITE ? "non-finite acceleration" : "");
    e->f16x2 = 0;
    *switched = 1;
  }
  return LB_OK;
}

extern "C" int lb_math_mode(lb_engine* e, int32_t set_mode, int32_t* mode_out, int32_t* flags_out) {
  if (!e) return lb_fail(LB_ERR_ARG, "null engine");
  if (set_mode < -1 || set_mode > 3)
    return lb_fail(LB_ERR_ARG, "set_mode must be -1 (query), 0 (f32), 1 (f16x2, guarded), 2 (f16x2, unguarded) or 3 (f16x2, "
                               "guarded, every tile of the batch kernels tested)");
  if (set_mode >= 0) {
    e->f16x2 = set_mode != 0;
    e->math_auto = set_mode == 1 || set_mode == 3;
    e->guard_full = set_mode == 3;
  }
  LB_HIP(hipMemcpyAsync(e->ctrl_host, e->ctrl, sizeof(lb_ctrl), hipMemcpyDeviceToHost, e->stream));
  LB_HIP(hipStreamSynchronize(e->stream));
  if (mode_out) *mode_out = e->f16x2 ? (e->math_auto ? (e->guard_full ? 3 : 1) : 2) : 0;
  if (flags_out) *flags_out = e->ctrl_host->math_flags;
  return LB_OK;
}

extern "C" int32_t lb_math_fallbacks(lb_engine* e) { return e ? e->math_fallbacks : 0; }

extern "C" int lb_debug_inject_guard(lb_engine* e, int32_t flags, int32_t step) {
  if (!e) return lb_fail(LB_ERR_ARG, "null engine");
  if (step >= 0 && !(flags & (LB_MATH_LARGE | LB_MATH_TINY | LB_MATH_NONFINITE)))
    return lb_fail(LB_ERR_ARG, "flags must carry at least one LB_MATH_* bit");
  //  only the guarded f16x2 mode consumes the hook: arming it in any other mode is refused instead of
  // leaving it armed for some later rollout
  if (step >= 0 && !(e->f16x2 && e->math_auto))
    return lb_fail(LB_ERR_STATE, "lb_debug_inject_guard needs the guarded f16x2 mode (lb_math_mode auto)");
  e->debug_guard_flags = flags;
  e->debug_guard_step = step < 0 ? -1 : step;
  return LB_OK;
}

extern "C" int lb_gns_forward(lb_engine* e, lb_gns* g, float* acc_out_dev) {
  if (!e || !g) return lb_fail(LB_ERR_ARG, "null argument");
  if (g->eng != e) return lb_fail(LB_ERR_ARG, "model was created for another engine");
  if (e->e_cap <= 0) return lb_fail(LB_ERR_STATE, "lb_gns_forward before lb_nl_allocate");
  if (e->g.force_kind == LB_FORCE_BUFFER && !e->force)
    return lb_fail(LB_ERR_STATE, "LB_FORCE_BUFFER engine: call lb_set_force first");
  LB_TRY(lbk_gns_forward(e, g));
  if (e->f16x2 && e->math_auto) {  // guarded mode: one host sync per stand-alone forward (not the rollout path)
    int switched = 0;
    LB_TRY(lb_math_check(e, &switched));
    if (switched) {  // this forward again in exact fp32; the engine returns to guarded f16x2 (round 4: not sticky)
      ++e->math_fallbacks;
      int rc = lbk_gns_forward(e, g);
      int sw2 = 0;
      if (!rc) rc = lb_math_check(e, &sw2);  // (engine on fp32: only clears what the fp32 decoder may have raised)
      e->f16x2 = 1;
      if (rc) return rc;
    }
  }
  if (acc_out_dev) {
    const int nb = (int)((e->BN + 255) / 256);
    hipLaunchKernelGGL(k_acc_export, dim3(nb), dim3(256), 0, e->stream, e->BN, e->g.dim, e->acc,
                       acc_out_dev);
    LB_HIP(hipGetLastError());
  }
  return LB_OK;
}

// --------------------------------------------------------------------- integrate / rollout
extern "C" int lb_integrate(lb_engine* e, const float* acc_dev, const double* target_dev,
                            double* pred_out_dev, int32_t pred_T) {
  if (!e || !target_dev) return lb_fail(LB_ERR_ARG, "null argument");
  const float* acc = acc_dev ? acc_dev : e->acc;
  const int stride = acc_dev ? e->g.dim : 4;
  return lbk_integrate(e, acc, stride, target_dev, nullptr, 0, pred_out_dev, pred_T);
}

extern "C" int lb_case_integrate(lb_engine* e, int32_t mode, const float* pred_dev,
                                 const double* pos_seq_dev, int32_t T, double* next_out_dev) {
  if (!e || !pred_dev || !pos_seq_dev || !next_out_dev) return lb_fail(LB_ERR_ARG, "null argument");
  if (mode != 0 && mode != 1) return lb_fail(LB_ERR_ARG, "mode must be 0 (acc) or 1 (vel)");
  if (T < 2 && mode == 0) return lb_fail(LB_ERR_ARG, "acc integration needs two frames");
  return lbk_case_integrate(e, mode, pred_dev, pos_seq_dev, T, next_out_dev);
}

static int gns_forward_thunk(lb_engine* e, void* model) { return lbk_gns_forward(e, (lb_gns*)model); }

extern "C" int lb_rollout(lb_engine* e, lb_gns* g, const double* traj_dev, int32_t T,
                          int32_t n_steps, double* pred_out_dev, int32_t* n_realloc_out) {
  if (!e || !g || !traj_dev || !pred_out_dev) return lb_fail(LB_ERR_ARG, "null argument");
  if (g->eng != e) return lb_fail(LB_ERR_ARG, "model was created for another engine");
  // the node-feature rows of every step ride along with its neighbor search
  struct FeatJob {
    lb_engine* e;
    ~FeatJob() { e->feat_job = lb_feat_job{}; }
  } feat_guard{e};
  LB_TRY(lb_gns_bind(e, g));
  e->feat_job = lb_feat_job{e->xnode, g->embed, g->desc.embedding_size, g->desc.num_particle_types, e->g.kpad, e->ptype, e->force};
  // the injection hook is one-shot per lb_rollout whatever the mode (a mode switch after arming must not leave it armed)
  const int inj_step = e->debug_guard_step, inj_flags = e->debug_guard_flags;
  e->debug_guard_step = -1;
  LB_TRY(lb_rollout_generic(e, gns_forward_thunk, g, traj_dev, T, n_steps, pred_out_dev, n_realloc_out));
  if (e->f16x2 && e->math_auto) {
    // The guard records the FIRST step at which a flag was raised (lb_ctrl::math_step).  Round 4: that ONE step is redone
    // in exact fp32 - window rebuilt from the input frames and the predictions already made - and the rollout continues
    // in f16x2 behind it (round 3 finished the rollout, and the engine's life, in fp32 at 2.3x the step time).  After
    // LB_GUARD_MAX_FALLBACKS (3) flagged steps the rest of THIS rollout runs in fp32; the next rollout starts in f16x2
    // again.  Steps before the flagged one are valid f16x2 work only if every tile was tested: with LB_GUARD=sampled
    // (first tile of every wave, rounds 2-3) the rollout is repeated from step 0 instead.
    static const int max_fb = getenv("LB_GUARD_MAX_FALLBACKS") ? atoi(getenv("LB_GUARD_MAX_FALLBACKS")) : 3;
    if (inj_step >= 0 && inj_step < n_steps) {
      // lb_debug_inject_guard (tests): behave as if the guard had ALSO fired at that step of THIS rollout - flags the
      // rollout raised itself stay (OR), the earlier step wins (min)
      int32_t cur[2] = {0, 0};
      LB_HIP(hipMemcpyAsync(&cur[0], &e->ctrl->math_flags, sizeof(int32_t), hipMemcpyDeviceToHost, e->stream));
      LB_HIP(hipMemcpyAsync(&cur[1], &e->ctrl->math_step, sizeof(int32_t), hipMemcpyDeviceToHost, e->stream));
      LB_HIP(hipStreamSynchronize(e->stream));
      const int32_t inj[2] = {cur[0] | inj_flags, cur[0] ? std::min(cur[1], (int32_t)inj_step) : (int32_t)inj_step};
      LB_HIP(hipMemcpyAsync(&e->ctrl->math_flags, &inj[0], sizeof(int32_t), hipMemcpyHostToDevice, e->stream));
      LB_HIP(hipMemcpyAsync(&e->ctrl->math_step, &inj[1], sizeof(int32_t), hipMemcpyHostToDevice, e->stream));
      LB_HIP(hipStreamSynchronize(e->stream));
    }
    int rc = LB_OK;
    for (int fallbacks = 0;;) {
      int switched = 0, s0 = 0;
      rc = lb_math_check(e, &switched, &s0);
      if (rc || !switched) break;   // (lb_math_check has put the engine on fp32)
      s0 = e->guard_sampled ? 0 : std::max(0, std::min(s0, n_steps - 1));
      int32_t n2 = 0;
      const bool rest = e->guard_sampled || ++fallbacks >= max_fb || s0 + 1 >= n_steps;
      e->math_fallbacks += rest ? n_steps - s0 : 1;
      rc = lb_rollout_generic(e, gns_forward_thunk, g, traj_dev, T, n_steps, pred_out_dev, &n2, s0, rest ? -1 : s0 + 1);
      if (n_realloc_out) *n_realloc_out += n2;
      if (rc) break;
      // Flags raised by the fp32 work itself (its decoder reports non-finite accelerations too) are final - there is no
      // better arithmetic to fall back to.  They are cleared here so that neither this loop nor the next call sees a
      // stale flag, and a step that flags again in fp32 is not retried: the rest of the rollout runs in fp32.
      int sw2 = 0, f32_flags = 0;
      rc = lb_math_check(e, &sw2, nullptr, &f32_flags);
      if (rc || rest) break;
      if (f32_flags) {
        e->math_fallbacks += n_steps - (s0 + 1);
        rc = lb_rollout_generic(e, gns_forward_thunk, g, traj_dev, T, n_steps, pred_out_dev, &n2, s0 + 1, -1);
        if (n_realloc_out) *n_realloc_out += n2;
        if (!rc) rc = lb_math_check(e, &sw2);
        break;
      }
      e->f16x2 = 1;                 // back to guarded f16x2 behind the flagged step
      rc = lb_rollout_generic(e, gns_forward_thunk, g, traj_dev, T, n_steps, pred_out_dev, &n2, s0 + 1);
      if (n_realloc_out) *n_realloc_out += n2;
      if (rc) break;
    }
    e->f16x2 = 1;                   // not sticky: the next rollout / forward starts in guarded f16x2
    if (rc) return rc;
  }
  return LB_OK;
}

// One rollout step (neighbor list -> model -> integrator) enqueued on e->stream.
static int lb_enqueue_step(lb_engine* e, int (*forward)(lb_engine*, void*), void* model,
                           const double* traj_dev, int32_t T, double* pred_out_dev, int32_t n_steps) {
  // two launches ride along with others in a rollout step (LB_SMALL_FUSED=0 switches that off): the node-feature rows
  // are written by extra workgroups of the neighbor search (lb_engine::feat_job, set by the model's rollout entry),
  // the integrator runs in the decoder's epilogue (lb_engine::integ_job)
  const bool fuse = lb_fused_launches();
  const lb_feat_job fj = e->feat_job;
  if (!fuse) e->feat_job.xnode = nullptr;
  e->feat_done = false;
  int rc = lbk_nl_build(e, false);
  e->feat_job = fj;
  e->integ_done = false;
  e->integ_job = lb_integ_job{e->win, e->blocks_done, e->ptype, traj_dev, T, pred_out_dev, n_steps, fuse ? 1 : 0};
  if (!rc) rc = forward(e, model);
  e->feat_done = false;
  e->integ_job.on = 0;
  const bool integrated = e->integ_done;
  e->integ_done = false;
  if (rc) return rc;
  if (integrated) return LB_OK;
  return lbk_integrate(e, e->acc, 4, nullptr, traj_dev, T, pred_out_dev, n_steps);
}

int lb_rollout_generic(lb_engine* e, int (*forward)(lb_engine*, void*), void* model,
                       const double* traj_dev, int32_t T, int32_t n_steps, double* pred_out_dev,
                       int32_t* n_realloc_out, int32_t start_step, int32_t stop_step) {
  if (T < e->g.isl) return lb_fail(LB_ERR_ARG, "trajectory shorter than input_seq_length");
  if (e->g.force_kind == LB_FORCE_BUFFER)
    return lb_fail(LB_ERR_UNSUPPORTED, "lb_rollout with LB_FORCE_BUFFER: drive the steps from the host");
  // LB_GRAPH=1: the step (about 50 launches whose arguments do not change between neighbor-list
  // re-allocations - the step index lives in the device control block) is captured once into a
  // hipGraph and replayed; pays off when the step is launch-bound (one small trajectory).
  // Capture needs a non-default stream: the rollout then runs on an engine-owned stream that is
  // ordered after the caller's stream on entry; lb_rollout is host-synchronous on exit anyway.
  static const bool want_graph = getenv("LB_GRAPH") && getenv("LB_GRAPH")[0] == '1';
  const bool use_graph = want_graph && !e->timers_on;
  hipStream_t user_stream = e->stream;
  hipGraphExec_t exec = nullptr;
  struct Restore {
    lb_engine* e;
    hipStream_t s;
    hipGraphExec_t* x;
    ~Restore() {
      if (*x) (void)hipGraphExecDestroy(*x);
      e->stream = s;
    }
  } restore{e, user_stream, &exec};
  if (use_graph) {
    if (!e->gstream) LB_HIP(hipStreamCreateWithFlags(&e->gstream, hipStreamNonBlocking));
    LB_HIP(hipEventRecord(e->step_ev[0], user_stream));
    LB_HIP(hipStreamWaitEvent(e->gstream, e->step_ev[0], 0));
    e->stream = e->gstream;
  }
  int n_realloc = 0, n_retry = 0;
  if (start_step > 0)   // resume: window of step start_step from the input frames + the predictions made so far
    LB_TRY(lbk_load_window_resume(e, traj_dev, T, pred_out_dev, n_steps, start_step));
  else
    LB_TRY(lbk_load_window(e, traj_dev, T, 0, 0));
  if (e->e_cap <= 0) LB_TRY(lb_nl_allocate(e, nullptr, nullptr, nullptr));
  int step = start_step;
  const int RA = 3;  // steps the host may run ahead of the device
  const int last = (stop_step >= 0 && stop_step < n_steps) ? stop_step : n_steps;  // steps [start_step, last) are run
  while (step < last) {
    bool warm = false;  // the first step after a (re-)allocation runs uncaptured: lazy buffer growth
    for (int s = step; s < last; ++s) {
      if (s - step >= RA) {
        // throttle: wait for step s-RA to retire, then look at the device-written host flag, so an
        // overflow costs at most RA steps of no-op launches instead of the rest of the rollout
        LB_HIP(hipEventSynchronize(e->step_ev[(s - RA) & 3]));
        if (*(volatile int32_t*)e->host_flag >= 0) break;
      }
      if (use_graph && warm && !exec) {
        hipGraph_t graph = nullptr;
        LB_HIP(hipStreamBeginCapture(e->stream, hipStreamCaptureModeThreadLocal));
        const int rc = lb_enqueue_step(e, forward, model, traj_dev, T, pred_out_dev, n_steps);
        const hipError_t ce = hipStreamEndCapture(e->stream, &graph);
        if (rc) return rc;
        if (ce != hipSuccess) return lb_fail(LB_ERR_HIP, "hipStreamEndCapture: %s", hipGetErrorString(ce));
        const hipError_t ie = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        if (ie != hipSuccess) return lb_fail(LB_ERR_HIP, "hipGraphInstantiate: %s", hipGetErrorString(ie));
      }
      if (exec)
        LB_HIP(hipGraphLaunch(exec, e->stream));
      else
        LB_TRY(lb_enqueue_step(e, forward, model, traj_dev, T, pred_out_dev, n_steps));
      warm = true;
      LB_HIP(hipEventRecord(e->step_ev[s & 3], e->stream));
    }
    LB_HIP(hipMemcpyAsync(e->ctrl_host, e->ctrl, sizeof(lb_ctrl), hipMemcpyDeviceToHost, e->stream));
    LB_HIP(hipStreamSynchronize(e->stream));
    if (e->ctrl_host->persist_error) {
      // A bounded spin of a single-launch neighbor build timed out - a busy or shared GPU can starve a predecessor
      // workgroup.  Everything from that step on is invalid, nothing before it:
      // the engine leaves that path and the rollout RESUMES at the recorded step on the multi-launch path (ADVICE r03;
      // round 3 surfaced LB_ERR_STATE here although the fall-back was one flag away).
      e->nl_one_off = true;
      const int s_bad = std::max(start_step, std::min(e->ctrl_host->persist_step, last - 1));
      const int32_t reset[2] = {0, LB_MATH_NO_STEP};
      LB_HIP(hipMemcpyAsync(&e->ctrl->persist_error, &reset[0], sizeof(int32_t), hipMemcpyHostToDevice, e->stream));
      LB_HIP(hipMemcpyAsync(&e->ctrl->persist_step, &reset[1], sizeof(int32_t), hipMemcpyHostToDevice, e->stream));
      // (... and may have raised the range guard on garbage latents: those flags are not real either)
      LB_HIP(hipMemcpyAsync(&e->ctrl->math_flags, &reset[0], sizeof(int32_t), hipMemcpyHostToDevice, e->stream));
      LB_HIP(hipMemcpyAsync(&e->ctrl->math_step, &reset[1], sizeof(int32_t), hipMemcpyHostToDevice, e->stream));
      // (the steps computed on the broken list may have poisoned the control block: that overflow is not real)
      const int32_t no_poison = -1;
      LB_HIP(hipMemcpyAsync(&e->ctrl->overflow_step, &no_poison, sizeof(int32_t), hipMemcpyHostToDevice, e->stream));
      e->host_flag[0] = -1;
      fprintf(stderr, "[lbhip] single-launch neighbor build gave up at step %d (spin time-out): resuming there on the multi-launch path\n",
              s_bad);
      if (exec) {
        (void)hipGraphExecDestroy(exec);
        exec = nullptr;
      }
      if (s_bad > 0)
        LB_TRY(lbk_load_window_resume(e, traj_dev, T, pred_out_dev, n_steps, s_bad));
      else
        LB_TRY(lbk_load_window(e, traj_dev, T, 0, 0));
      step = s_bad;
      if (++n_retry > 4) return lb_fail(LB_ERR_STATE, "single-launch paths keep timing out");
      continue;
    }
    LB_TRY(lb_check_density(e));
    if (e->ctrl_host->overflow_step < 0) break;
    // (eval) Reallocate neighbors list at step k - rollout.py:139-151.  Every kernel after the
    // overflowing build was a no-op, so window and step counter still describe step k.
    step = e->ctrl_host->step;
    ++n_realloc;
    if (n_realloc > n_steps + 8) return lb_fail(LB_ERR_STATE, "neighbor list keeps overflowing");
    if (exec) {  // capacities, buffers and kernel variants may change: capture again
      (void)hipGraphExecDestroy(exec);
      exec = nullptr;
    }
    LB_TRY(lb_nl_allocate(e, nullptr, nullptr, nullptr));
  }
  if (n_realloc_out) *n_realloc_out = n_realloc;
  return LB_OK;
}

extern "C" int lb_ekin(lb_engine* e, const double* rollout_dev, int32_t T, int32_t stride, double dt,
                       double dx, double* out_dev, int32_t n_out) {
  if (!e || !rollout_dev || !out_dev) return lb_fail(LB_ERR_ARG, "null argument");
  if (stride < 1 || T < 2) return lb_fail(LB_ERR_ARG, "need stride >= 1 and T >= 2");
  const int expect = (T - 1 + stride - 1) / stride;  // len(x[1::stride]) == len(x[0:-1:stride])
  if (n_out != expect) return lb_fail(LB_ERR_ARG, "n_out must be ceil((T-1)/stride) = %d", expect);
  return lbk_ekin(e, rollout_dev, T, stride, n_out, dt, dx, out_dev);
}

extern "C" int lb_sinkhorn(lb_engine* e, const double* pred_dev, int32_t pred_T, const double* target_dev,
                           int32_t target_T, int32_t stride, double threshold, double* out_dev, int32_t n_out,
                           int32_t* iters_out_host) {
  if (!e || !pred_dev || !target_dev || !out_dev) return lb_fail(LB_ERR_ARG, "null argument");
  if (stride < 1 || pred_T < 1 || target_T < 1) return lb_fail(LB_ERR_ARG, "need stride >= 1 and T >= 1");
  if (!(threshold > 0)) return lb_fail(LB_ERR_ARG, "threshold must be > 0");
  const int T = pred_T < target_T ? pred_T : target_T;
  const int expect = (T + stride - 1) / stride;  // len(x[0::stride])
  if (n_out != expect) return lb_fail(LB_ERR_ARG, "n_out must be ceil(T/stride) = %d", expect);
  return lbk_sinkhorn(e, pred_dev, pred_T, target_dev, target_T, stride, n_out, threshold, out_dev, iters_out_host);
}

extern "C" int lb_sinkhorn_pot(lb_engine* e, const double* pred_dev, int32_t pred_T, const double* target_dev,
                               int32_t target_T, int32_t stride, double reg, int32_t num_iter_max, double stop_thr,
                               double* out_dev, int32_t n_out, int32_t* info_out_host) {
  if (!e || !pred_dev || !target_dev || !out_dev) return lb_fail(LB_ERR_ARG, "null argument");
  if (stride < 1 || pred_T < 1 || target_T < 1) return lb_fail(LB_ERR_ARG, "need stride >= 1 and T >= 1");
  if (!(reg > 0) || num_iter_max < 1 || !(stop_thr >= 0)) return lb_fail(LB_ERR_ARG, "need reg > 0, numItermax >= 1, stopThr >= 0");
  const int T = pred_T < target_T ? pred_T : target_T;
  const int expect = (T + stride - 1) / stride;  // len(x[0::stride])
  if (n_out != expect) return lb_fail(LB_ERR_ARG, "n_out must be ceil(T/stride) = %d", expect);
  return lbk_sinkhorn_pot(e, pred_dev, pred_T, target_dev, target_T, stride, n_out, reg, num_iter_max, stop_thr, out_dev,
                          info_out_host);
}

extern "C" int lb_metrics(lb_engine* e, const double* pred_dev, int32_t pred_T,
                          const double* target_dev, int32_t target_T, int32_t n_steps, double* mse,
                          double* mae) {
  if (!e || !pred_dev || !target_dev) return lb_fail(LB_ERR_ARG, "null argument");
  if (n_steps > pred_T || n_steps > target_T) return lb_fail(LB_ERR_ARG, "n_steps exceeds pred_T/target_T");
  return lbk_metrics(e, pred_dev, pred_T, target_dev, target_T, n_steps, mse, mae);
}
