// lb_internal.h - engine object, device control block and kernel launchers shared by the
// translation units of liblbhip.so.  gfx950 only (wave64, MFMA f32 32x32x2).
#pragma once
#include <cstdlib>
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/lbhip.h"

#define LB_MAX_STENCIL_CAND 2048  // candidates staged in LDS per cell block (27*cap must fit)
#define LB_MAX_ROW 256            // row buffer of the default neighbor kernels (neighbors per particle)
#define LB_MAX_ROW_DENSE 4096     // ... of the dense fall-back (wave-per-receiver kernel, dynamic LDS)
#define LB_TILE 32                // rows (edges / nodes) per wave tile: the N of the 32x32x2 MFMA
#define LB_D 128                  // latent width built so far

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---------------------------------------------------------------------------------------------
// Device-resident control block: everything a kernel needs to size itself without a host sync.
struct lb_ctrl {
  int32_t step;            // rollout step counter (ring head = step % isl)
  int32_t overflow_step;   // first step whose neighbor list overflowed, -1 = none ("poison")
  int32_t n_edges_total;   // real edges over all B trajectories, clamped to B*E_cap
  int32_t n_edges_unclamped;
  int32_t max_cell_occ;    // max particles in one cell (this build)
  int32_t density_error;   // stencil/row exceeded the LDS tile bounds
  int32_t max_deg;         // max receiver degree (this build)
  int32_t row_overflow;    // a row outgrew the per-node slots of the single-sweep update path
  int32_t math_flags;      // f16x2 range guard: 1 = an operand >= 2^15 (fp16 overflow in reach), 2 = a
                           // whole operand tile < 2^-10 (the fp16 `lo` halves go subnormal),
                           // 4 = non-finite accelerations; checked by the host at its sync points
  float ln_inv_d;          // LayerNorm over a latent narrower than the 128-wide tiles (zero-padded weights):
  float ln_pad;            // mean = sum / d, var = (sum_128 (x-mean)^2 - pad * mean^2) / d, pad = 128 - d
  int32_t persist_error;   // 2: a single-launch neighbor build (k_nl_small / k_nl_mid) gave up a bounded spin; results are invalid
  int32_t nl_epoch;        // k_nl_small: build counter that tags the per-workgroup edge counts (never 0 mod 2^16)
  int32_t persist_step;    // first rollout step at which persist_error was raised (0x7fffffff = none): lb_rollout resumes THERE on
                           // the multi-launch path
  int32_t math_step;       // first rollout step at which a range-guard flag was raised (0x7fffffff = none): lb_rollout
                           // resumes THERE in exact fp32 instead of repeating the rollout
  int32_t acct_builds;     // lb_edge_accounting: neighbor-list builds since the last reset,
  int32_t acct_first;      // the first of them's edge count,
  int64_t acct_sum;        // and the sum of their (unclamped) edge counts
};
// every finisher of a neighbor-list build calls this from ONE thread, next to its n_edges_unclamped store
__device__ __forceinline__ void lb_acct_edges(lb_ctrl* ctrl, int total) {
  if (ctrl->acct_builds == 0) ctrl->acct_first = total;
  ctrl->acct_builds += 1;
  ctrl->acct_sum += total;
}
#define LB_MATH_NO_STEP 0x7fffffff
#define LB_MATH_LARGE 1
#define LB_MATH_TINY 2
#define LB_MATH_NONFINITE 4

// Geometry + normalisation constants, passed by value to kernels.
struct lb_geom {
  int32_t dim, N, B, isl, periodic, has_bound, has_vel_mag, force_kind, force_axis;
  int32_t use_cell_list;   // jax-md: cutoff < box/3 in every dim, else all-pairs candidates
  int32_t ncell[3];        // cells per side (1 when !use_cell_list)
  int32_t ncells;          // cells per trajectory
  int32_t nstencil;        // 3^dim, or 1 when !use_cell_list
  int32_t node_in;         // feature columns without embedding
  int32_t kpad;            // node feature row stride (multiple of 32)
  int32_t f32;             // dtype=float32 geometry: every arithmetic result rounded to float (lb_device.h: lb_r)
  double cell_size[3];     // f32-rounded like jax-md, widened
  double box[3], half_box[3];
  double rc, rc2;
  double vel_mean[3], vel_std[3], acc_mean[3], acc_std[3];
  double bound_lo[3], bound_hi[3];
  double force_split, force_lo[3], force_hi[3];
};

// node features of ONE rollout step riding along with the neighbor search (lb_engine::feat_job)
struct lb_feat_job {
  float* xnode;        // [BN][kpad] network input rows (null = no job)
  const float* embed;  // particle-type embedding table or null
  int32_t emb, ntypes, kpad;
  const int32_t* ptype;
  const double* force;
};

// integrator of ONE rollout step riding along with the decoder (lb_engine::integ_job)
struct lb_integ_job {
  double* win;          // null = no job
  int32_t* blocks_done;
  const int32_t* ptype;
  const double* traj;
  int32_t T;
  double* pred;
  int32_t pred_T;
  int32_t on;
};

enum lb_timer_class {
  LB_T_CELLS = 0,    // cell binning: count + scan + fill
  LB_T_NEIGH,        // stencil search: count pass + row scan + fill pass (+edge features)
  LB_T_NODEFEAT,     // node feature assembly
  LB_T_ENC_NODE,     // encoder node MLP (+ first projection)
  LB_T_ENC_EDGE,     // encoder edge MLP
  LB_T_EDGE_MLP,     // processor edge MLP (gather + 2 GEMM + LN + residual)
  LB_T_AGGREGATE,    // segment_sum over receivers
  LB_T_NODE_MLP,     // processor node MLP (+ next projection)
  LB_T_DECODER,      // decoder MLP
  LB_T_INTEGRATE,    // integrator + kinematic select + window shift + prediction store
  LB_T_MISC,
  LB_T_PROCESSOR,    // all message-passing layers as one persistent launch (lb_persist.hip)
  LB_T_EDGE_LAST,    // the LAST processor layer's edge MLP when its updated edge latents have no reader (no store:
                     // half the HBM bytes of the other layers' launches - timed apart so that the roofline of
                     // LB_T_EDGE_MLP is not credited bytes this variant does not move)
  LB_T_COUNT
};

struct lb_timer_rec {
  hipEvent_t a, b;
  int cls;
};

struct lb_engine {
  lb_case_desc desc;
  lb_geom g;
  hipStream_t stream;
  int64_t BN;

  // state
  double* win;        // [isl][dim][B*N] SoA ring of positions
  int32_t* ptype;     // [B*N]
  double* force;      // [B*N*dim] (LB_FORCE_BUFFER) or null
  lb_ctrl* ctrl;      // device
  int32_t* blocks_done = nullptr;  // k_integrate: workgroups that have read the step counter (the last one advances it)
  lb_ctrl* ctrl_host; // pinned mirror
  int32_t* host_flag;     // pinned + device-mapped: first overflowing step (-1 = none), written by
  int32_t* host_flag_dev; //   k_row_scan so lb_rollout can stop enqueuing without a stream sync
  hipEvent_t step_ev[4];  // run-ahead throttle of lb_rollout
  hipStream_t gstream;    // engine-owned stream of the hipGraph rollout (LB_GRAPH=1), lazily created

  // neighbor structures
  int32_t cell_capacity, e_cap;      // frozen capacities (0 = not allocated)
  int64_t e_alloc;                   // allocated edge slots (>= B*e_cap)
  int32_t* cell_of;    // [BN] global cell id of each particle
  int32_t* cell_count; // [B*ncells]
  int32_t* cell_start; // [B*ncells+1]
  int32_t* cell_fill;  // [B*ncells]
  bool cells_traj_ready = false;  // k_cells_traj: cell_count[0 .. B] hold its occupancy slots + arrival ticket, zeroed
  int32_t* cell_part;  // [BN] particle ids grouped by cell
  int32_t* deg;        // [BN]
  lb_feat_job feat_job{};    // rollout step: node features ride along with the neighbor search (xnode == null: no job)
  bool feat_done = false;    // ... and were written by it: the model's forward skips its own feature launch
  lb_integ_job integ_job{};  // rollout step: the integrator rides along with the decoder (win == null: no job)
  bool integ_done = false;
  bool nl_one_off = false;        // k_nl_small timed out once: multi-launch build from now on
  unsigned long long* nl_wg_sum = nullptr;  // [BN / 8 + 2] k_nl_small: epoch-tagged words of the workgroups
  int32_t* row_ptr;    // [BN+1]
  int32_t* scan_part;  // partial sums of the two-level scans
  double* cpos;        // [dim][cell_slots] newest-frame positions in cell-sorted order (CSR: BN slots; fixed-stride: cells * cap)
  int64_t cell_slots = 0;      // slots allocated for cell_part / cpos (>= BN)
  bool cells_strided = false;  // this build binned into fixed-stride slots (lb_neighbor.hip: k_cell_bin)
  bool nl_dense = false;  // sticky: the density exceeded the staged kernel's limits once -> wave-per-receiver kernel
  int32_t row_cap = 0;    // its LDS row buffer (entries per wave), sized from the largest degree seen
  int32_t maxd;        // per-node slot count of the single-sweep update path (0 = not sized)
  int32_t* tmp_send;   // [BN][maxd] sorted sender rows before compaction
  float* tmp_feat;     // [BN][maxd][4]
  double* tmp_feat64;  // [BN][maxd][4]
  int32_t* senders;    // [e_alloc] global node ids
  int32_t* receivers;  // [e_alloc]
  float* efeat;        // [e_alloc][8]  rel_disp(dim), rel_dist, zero pad
  double* efeat64;     // [e_alloc][4]  fp64 copy for the API (allocated on demand)
  int32_t* overflow;   // [B] did_buffer_overflow
  int32_t* nedges_b;   // [B]

  // network scratch (allocated by lb_gns_create / regrown with e_alloc)
  float* xnode;        // [BN][kpad]
  float* nlat;         // [BN][D]
  float* agg;          // [BN][D]
  float* psr;          // [BN][2D]  projections of the node latents for the edge MLP
  float* elat;         // [e_alloc][D]
  float* msg;          // [e_alloc][D]   (stand-alone segment_sum path only)
  float* part;         // [e_alloc/16+2][2][D] partial sums of receivers cut by a tile boundary: inside the agg allocation
  int64_t aggpart_bytes = 0;  // bytes of that allocation (agg | part)
  int fused_agg;       // 1: aggregation fused into the edge kernel (default), 0: msg + k_segment_sum
  int f16x2;           // 1: GEMMs in fp16 hi/lo split arithmetic on the fp16 MFMA (fp32-class accuracy)
  int math_auto;       // 1: f16x2 with the range guard - a raised lb_ctrl::math_flags makes the host repeat
                       //    the work in exact-fp32 MFMA arithmetic and stay there (LB_MATH unset);
                       // 0: the mode LB_MATH / lb_math_mode fixed
  int guard_sampled;   // 1 (LB_GUARD=sampled): only the sampled probe of rounds 2-3 (first tile of every wave); a raised flag then
                       // repeats the rollout from step 0 (earlier steps may have had unsampled out-of-range tiles)
  int math_fallbacks = 0;  // steps redone in exact fp32 by the guard since the engine was created (lb_stats)
  int debug_guard_step = -1;   // lb_debug_inject_guard: the next lb_rollout behaves as if the guard fired at this step
  int debug_guard_flags = 0;
  int guard_full;      // 1: the wave-per-tile edge kernel tests EVERY tile for the TINY condition (lb_math_mode 3 /
                       //    LB_GUARD=full; +8 % on that kernel); 0: sampled probe.  The M-split kernels always test all.
  float* acc;          // [BN][4] decoder output (dim padded to 4)
  const void* bound_model = nullptr;  // the lb_gns whose per-model constants (LayerNorm width, node row stride) are
                                      // currently in the control block / geometry: lb_gns_bind

  // timers
  bool timers_on;
  std::vector<lb_timer_rec> trecs;
  bool ext_armed = false, ext_used = false;  // single-kernel timer classes: events bound to the dispatch itself
  std::vector<hipEvent_t> epool;
  double t_ms[LB_T_COUNT];
  int64_t t_n[LB_T_COUNT];
};

struct lb_mlp_w {      // one packed 2-layer MLP on the device
  const float* w0;     // packed fragments, K0pad x D
  const float* b0;     // [D]
  const float* w1;     // packed, D x Mpad
  const float* b1;     // [Mpad]
  const float* ln_s;   // [D] or null
  const float* ln_o;
};

struct lb_edge16_args {  // lb_edge16.hip
  const lb_ctrl* ctrl;
  const int32_t* senders;
  const int32_t* receivers;
  const float* efeat;  // ENC input [E][8]
  float* elat;         // [E][128] in/out
  float* elat_out;     // where the updated latents go (null = in place)
  float* msg;          // [E][128] out (PROC, !fused)
  const float* psr;    // [BN][256]
  const float* w0p;    // 16-packed: PROC 128x128 (edge rows of W0), ENC 16x128
  const float* b0;     // ENC only
  const float* w1p;    // 16-packed 128x128
  const float* b1;
  const float* ln_s;
  const float* ln_o;
  int fused;
  const int32_t* row_ptr;
  float* agg;
  float* part;         // [ceil(E/16)][2][128]; = agg + BN*128 (one allocation of aggpart_bytes bytes)
  int64_t aggpart_bytes;
  int reverse;         // k_edge16w: every XCD walks its range of tiles from the end (odd layers: the latents the layer before
                       // wrote last are still in the 256 MiB Infinity Cache)
  int skip_elat_store; // last processor layer: the updated edge latents have no reader
};

struct lb_node_args {
  const lb_ctrl* ctrl;
  int64_t n_rows;
  const float* xin;   // ENC: [rows][8*NKQ_A]; PROC: nlat [rows][128]
  const float* agg;   // PROC: [rows][128]
  float* nlat;        // out [rows][128]
  const float* w0p;   // packed (8*(NKQ_A+NKQ_B)) x 128
  const float* b0;
  const float* w1p;
  const float* b1;
  const float* ln_s;
  const float* ln_o;
  const float* wpp;   // packed 128 x 256 projection for the NEXT edge MLP, or null
  const float* bp;    // [256]
  float* psr;         // out [rows][256]
  int fused;          // agg comes from the fused edge epilogue (agg + per-tile partial slots)
  int tile_shift;     // log2 of the edge kernel's tile (4 or 5)
  const int32_t* row_ptr;
  const float* part;
};

struct lb_gen_lin {                 // one Linear of the generic path
  std::vector<const float*> wh, wf;  // per 128-row input block: f16x2 (hi|lo) and fp32 fragment packings
  const float* b;                    // [128]
};
struct lb_gen_mlp {
  std::vector<lb_gen_lin> lin;
  const float* ln_s = nullptr;
  const float* ln_o = nullptr;
};

struct lb_gns {
  lb_gns_desc desc;
  lb_engine* eng;
  float* blob;         // single device allocation holding everything below
  const float* embed;  // [types][emb]
  lb_mlp_w enc_node, enc_edge, dec;
  std::vector<lb_mlp_w> proc_edge, proc_node;  // proc_edge[k].w0 packs only the edge-latent rows
  std::vector<const float*> proj_w;            // packed [D x 2D]: sender | receiver rows of w0
  std::vector<const float*> proj_b;            // [2D]: zeros | b0
  const float* enc_edge_w0_16;                 // 16-row-tile packings of the edge MLP matrices
  const float* enc_edge_w1_16;
  std::vector<const float*> proc_edge_w0_16, proc_edge_w1_16;
  const float* enc_edge_w0_16h;                // f16x2 (hi|lo) packings
  const float* enc_edge_w1_16h;
  std::vector<const float*> proc_edge_w0_16h, proc_edge_w1_16h;
  // f16x2 node-MLP packings: w0 (Kpad x 128), w1 (128 x 128), projection (128 x 256)
  const float* enc_node_w0_h;
  const float* enc_node_w1_h;
  std::vector<const float*> proc_node_w0_h, proc_node_w1_h, proj_w_h;
  std::vector<const float*> proj_w_h2;  // projection packed as two 128-wide halves [Ws | Wr] (lb_node16s.hip)
  int kq_node;         // node_in(+emb) padded to a multiple of 32, in units of 8
  float lnc[2] = {1.0f / LB_D, 0.f};  // lb_ctrl::ln_inv_d, ln_pad of this model (latent width < 128: zero padded)
  float* tap;
  // decoder on the 16-row tile scheme (k_decoder16): W0 and the out_dim block of W1, both packings
  const float* dec_w0_h = nullptr;
  const float* dec_w0_f = nullptr;
  const float* dec_w1_h = nullptr;
  const float* dec_w1_f = nullptr;
  float dec_unscale = 1.f;  // the f16x2 copy of the decoder's output Linear is packed times a power of two
  // M-split images (lb_msplit.hip, small graphs): per MLP [W0 | W1 (| projection of the NEXT edge MLP)]
  const float* ms_enc_node = nullptr;
  const float* ms_enc_edge = nullptr;
  std::vector<const float*> ms_proc_edge, ms_proc_node;
  // num_mlp_layers != 2 (lb_gns_generic.hip): one packed 128x128 Linear per input block, both packings
  bool generic = false;
  lb_gen_mlp g_enc_node, g_enc_edge, g_dec;
  std::vector<lb_gen_mlp> g_proc_edge, g_proc_node;
  float* gen_hn[3] = {nullptr, nullptr, nullptr};  // node-sized hidden / projection scratch
  float* gen_he = nullptr;                          // edge-sized hidden scratch (tile-blocked)
  int64_t gen_he_cap = 0;
};

// ---------------------------------------------------------------------------------------------
extern thread_local std::string g_lb_err;
int lb_fail(int code, const char* fmt, ...);
#define LB_HIP(call)                                                                   \
  do {                                                                                 \
    hipError_t _e = (call);                                                            \
    if (_e != hipSuccess)                                                              \
      return lb_fail(LB_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(_e), \
                     __FILE__, __LINE__);                                              \
  } while (0)

template <typename T>
static inline int lb_alloc(T** p, size_t n) {
  *p = nullptr;
  if (n == 0) n = 1;
  LB_HIP(hipMalloc((void**)p, n * sizeof(T)));
  return LB_OK;
}
#define LB_TRY(x)          \
  do {                     \
    int _rc = (x);         \
    if (_rc) return _rc;   \
  } while (0)

// LB_SMALL_FUSED=0 (one switch since round 6; tests/test_switches_gpu.py): every launch-saving fusion off - the single-launch
// cell binning / neighbor builds / degree scan + compaction, node features and integrator riding with other launches, the
// decoder inside the last M-split node launch - i.e. the general multi-launch paths that big or dense problems take anyway.
static inline bool lb_fused_launches() {
  static const bool ok = !(getenv("LB_SMALL_FUSED") && getenv("LB_SMALL_FUSED")[0] == '0');
  return ok;
}
void lb_tic(lb_engine* e, int cls);
void lb_toc(lb_engine* e);
// Timer classes that consist of ONE kernel (processor edge / node MLP): between lb_tic_single and lb_toc
// the launcher uses LB_LAUNCH_TIMED, which binds the class's two events to the dispatch itself
// (hipExtLaunchKernelGGL start/stop events), so the elapsed time is the kernel's own begin -> end - the
// number rocprofv3 --kernel-trace reports - without the dispatch gap and marker packets that a
// record / launch / record bracket adds (~15-20 us per launch).
void lb_tic_single(lb_engine* e, int cls);
#define LB_LAUNCH_TIMED(e, kern, grid, block, ...)                                                        \
  do {                                                                                                    \
    if ((e)->ext_armed && !(e)->ext_used) {                                                               \
      hipExtLaunchKernelGGL(kern, grid, block, 0, (e)->stream, (e)->trecs.back().a, (e)->trecs.back().b, \
                            0, __VA_ARGS__);                                                              \
      (e)->ext_used = true;                                                                               \
    } else {                                                                                              \
      hipLaunchKernelGGL(kern, grid, block, 0, (e)->stream, __VA_ARGS__);                                 \
    }                                                                                                     \
  } while (0)

// lb_api.hip
int lb_ensure_edges(lb_engine* e, int64_t need);
int lb_alloc_aggpart(lb_engine* e);

// lb_neighbor.hip
int lbk_nl_build(lb_engine* e, bool want_efeat64);
int lbk_nl_export(lb_engine* e, int32_t* idx_out, int32_t* n_edges_out);
int lbk_edge_features_export(lb_engine* e, double* rel_disp, double* rel_dist);

// lb_state.hip
int lbk_load_window(lb_engine* e, const double* traj, int T, int t0, int step);
int lbk_load_window_resume(lb_engine* e, const double* traj, int T, const double* pred, int pred_T, int step);
int lbk_read_window(lb_engine* e, double* out);
int lbk_node_features(lb_engine* e, float* xnode, const float* embed, int emb, int ntypes,
                      double* vel_hist, double* vel_mag, double* bound, double* force);
int lbk_integrate(lb_engine* e, const float* acc, int acc_stride, const double* target,
                  const double* traj, int T, double* pred, int pred_T);
int lbk_ekin(lb_engine* e, const double* roll, int T, int stride, int n_out, double dt, double dx,
             double* out);
int lbk_case_integrate(lb_engine* e, int mode, const float* pred, const double* pos_seq, int T,
                       double* out);
int lbk_metrics(lb_engine* e, const double* pred, int pred_T, const double* target, int target_T,
                int n_steps, double* mse, double* mae);

// lb_sinkhorn.hip
int lbk_sinkhorn(lb_engine* e, const double* pred, int pred_T, const double* target, int target_T, int stride,
                 int n_out, double threshold, double* out_dev, int32_t* iters_host);
int lbk_sinkhorn_pot(lb_engine* e, const double* pred, int pred_T, const double* target, int target_T, int stride,
                     int n_out, double reg, int max_it, double thr, double* out_dev, int32_t* info_host);

int lbk_node_features_raw(lb_engine* e, float* xnode, int kpad);

// lb_api.hip: the device-resident step loop shared by the models
int lb_rollout_generic(lb_engine* e, int (*forward)(lb_engine*, void*), void* model,
                       const double* traj_dev, int32_t T, int32_t n_steps, double* pred_out_dev,
                       int32_t* n_realloc_out, int32_t start_step = 0, int32_t stop_step = -1);

// lb_segnn.hip
struct lb_segnn;
int lbk_segnn_forward(lb_engine* e, lb_segnn* m);
int lbk_sg_prep(lb_engine* e, int homogeneous, int vel_avg, int ns4, int nv4, float* xnode, float* eattr, float* msgsv,
                float* nodesv, float* nattr, int64_t ecap);

// lb_segnn_gen.hip: general irreps (lmax <= 2) and the e3nn BatchNorm switches
struct lb_sgg;
int lb_sgg_create(lb_engine* e, const lb_segnn_desc* d, const float* w, int64_t n_floats, lb_sgg** out);
void lb_sgg_destroy(lb_sgg* m);
void lb_sgg_set_tap(lb_sgg* m, float* tap);
int lb_sgg_row_floats(const lb_sgg* m);
int lbk_sgg_forward(lb_engine* e, lb_sgg* m);

// lb_segnn_msg.hip
void lb_sg_msg_image(const float* ws0, const float* wv0, const float* b0, const float* ws1,
                     const float* wv1, const float* b1, float* out);
int lb_sg_msg_image_floats(void);
int lbk_sg_message(lb_engine* e, const float* f, const float* image, float* agg, float* part, int64_t out_bytes,
                   bool finish);
void lb_sg_upd_image(const float* ws0, const float* wv0, const float* b0, const float* ws1,
                     const float* wv1, const float* b1, float* out);
int lb_sg_upd_image_floats(void);
int lbk_sg_update(lb_engine* e, float* f, const float* agg, const float* part, const float* nattr, const float* image,
                  bool combine_partials);

// lb_segnn_node.hip
void lb_sg_embed_image(const float* ws, const float* wv, const float* b, int ns, int nv, float* out);
int lb_sg_embed_image_floats(void);
int lbk_sg_embed(lb_engine* e, const float* xnode, const float* image, float* f, float* nattr, int homogeneous,
                 int vel_avg);
void lb_sg_readout_image(const float* ws0, const float* wv0, const float* b0, const float* ws1, const float* wv1,
                         const float* b1, const float* wo, float* out);
int lb_sg_readout_image_floats(void);
int lbk_sg_readout(lb_engine* e, const float* f, const float* nattr, const float* image, float* acc_out);

// lb_api.hip: node-sized network scratch (allocated once per engine, xnode at the full 128-column width) and the
// per-model constants that live in engine-wide state; several models may share one engine
int lb_ensure_node_scratch(lb_engine* e);
int lb_gns_bind(lb_engine* e, lb_gns* g);
// lb_gns.hip
int lbk_gns_forward(lb_engine* e, lb_gns* g);
// lb_gns_generic.hip: num_mlp_layers != 2
int lb_gns_create_generic(lb_engine* e, const lb_gns_desc* d, const float* w, int64_t n_floats, lb_gns** out);
int lbk_gns_forward_generic(lb_engine* e, lb_gns* g);
int lbk_decoder16(lb_engine* e, lb_gns* g);
int lbk_segment_sum(lb_engine* e, const float* msg, float* out, int D);
void lb_pack_weight(const float* w, int K, int M, int Kpad, int Mpad, float* out);

// lb_edge16.hip
void lb_pack_weight16(const float* w, int K, int M, int Kpad, float* out);
void lb_pack_weight16h(const float* w, int K, int M, int Kpad, float* out, int Mpad = 128);

// lb_node16s.hip: round-2 node kernel (one weight pass per CU through a direct-to-LDS ring);
// wph2 = projection packed as [Ws | Wr] halves, or null
int lbk_node16s(lb_engine* e, const lb_node_args& a, const float* w0h, const float* w1h,
                const float* wph2, int npa, int npb, bool resid);
int lbk_edge16(lb_engine* e, const lb_edge16_args& a, bool proc, bool f16x2);
// lb_edge16v.hip: processor edge kernel (f16x2, fused aggregation, two waves per SIMD)
int lbk_edge16v(lb_engine* e, const lb_edge16_args& a);
// lb_edge16w.hip: the same kernel with the deferred epilogue (round 5); bit-identical results
int lbk_edge16w(lb_engine* e, const lb_edge16_args& a);
int lbk_edge_enc16v(lb_engine* e, const lb_edge16_args& a);
