/*
 * lbhip.h - C ABI of liblbhip.so: the MI355X (gfx950) rollout engine that sits behind
 * LagrangeBench's case_setup / models.GNS / evaluate.rollout Python API.
 *
 * The reference (tumaer/lagrangebench) is pure Python/JAX and has no FFI of its own; each
 * entry point below replaces the reference function named in its comment (paths relative to
 * the reference repo root).  A maintainer binds them with ctypes - see INTEGRATION.md.
 *
 * Conventions
 *   - every pointer suffixed _dev is DEVICE memory owned by the caller (e.g. a torch tensor);
 *     the library owns only its opaque handles and internal scratch.
 *   - all work is enqueued on the hipStream_t given to lb_engine_create(); only the calls
 *     documented as "host-synchronous" block.
 *   - return value: 0 = LB_OK, < 0 = error (lb_strerror()).  Neighbor-list overflow is NOT an
 *     error: like the reference (evaluate/rollout.py:134-151) it is a flag the driver polls.
 *   - positions/targets are fp64 (the reference's default dtype, defaults.py:22); network
 *     inputs/outputs are fp32 (runner.py:71-72).  Indices are int32.
 *   - B independent trajectories ("batch", the reference's vmap axis rollout.py:226-228) of
 *     N particles each are processed as one disjoint graph of B*N nodes.
 */
#ifndef LBHIP_H
#define LBHIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LB_OK 0
#define LB_ERR_ARG (-1)         /* bad argument / unsupported configuration            */
#define LB_ERR_HIP (-2)         /* a HIP runtime call failed (message via lb_strerror) */
#define LB_ERR_STATE (-3)       /* call order violated (e.g. update before allocate)   */
#define LB_ERR_DENSITY (-4)     /* a cell stencil / row exceeds the LDS tile bounds    */
#define LB_ERR_UNSUPPORTED (-5) /* valid in the reference, not built yet               */

#define LB_FORCE_NONE 0
#define LB_FORCE_PIECEWISE 1 /* f = pos[axis] > split ? f_hi : f_lo (RPF body force, DAM gravity) */
#define LB_FORCE_BUFFER 2    /* caller supplies (B,N,dim) fp64 every step via lb_set_force()       */

typedef struct lb_engine lb_engine;
typedef struct lb_gns lb_gns;

/* What case_builder(box, metadata, input_seq_length, cfg_neighbors, cfg_model, noise_std,
 * external_force_fn, dtype) closes over: case_setup/case.py:62-140.  Normalisation stats are the
 * OUTPUT of get_dataset_stats (data/utils.py:9-45), i.e. noise already folded into std. */
typedef struct lb_case_desc {
  int32_t dim;          /* 2 or 3 */
  int32_t n_particles;  /* N per trajectory (metadata["num_particles_max"]) */
  int32_t batch;        /* B trajectories advanced together */
  int32_t isl;          /* input_seq_length (>= 2) */
  int32_t periodic;     /* any(metadata["periodic_boundary_conditions"]) - case.py:104-108 */
  int32_t has_bound;    /* "bound" node feature: not any(pbc) - features.py:87-103 */
  int32_t has_vel_mag;  /* cfg_model.magnitude_features - features.py:80-85 */
  int32_t force_kind;   /* LB_FORCE_* : external_force_fn - features.py:105-107 */
  int32_t force_axis;
  int32_t geometry_f32; /* dtype of case_builder (case.py:169): 0 float64 (default), 1 float32 - buffers stay fp64, every
                         * value is float-representable and every arithmetic result is rounded to float */
  double box[3];
  double r_cutoff;      /* metadata["default_connectivity_radius"] */
  double capacity_multiplier; /* cfg_neighbors.multiplier */
  double vel_mean[3], vel_std[3], acc_mean[3], acc_std[3];
  double bound_lo[3], bound_hi[3]; /* metadata["bounds"] */
  double force_split;
  double force_lo[3], force_hi[3];
} lb_case_desc;

/* GNS hyper-parameters: models/gns.py:36-63 (runner.py:205-216). */
typedef struct lb_gns_desc {
  int32_t latent_size;        /* 128 (GNS-10-128); any multiple of 16 up to 128 (GNS-5-64) runs zero-padded */
  int32_t blocks_per_step;    /* num_mlp_layers >= 1 (models/utils.py:100-115); 2 = the fused kernels, other depths
                               * run one Linear per launch (csrc/lb_gns_generic.hip) */
  int32_t num_mp_steps;
  int32_t embedding_size;     /* particle_type_embedding_size (16) */
  int32_t num_particle_types; /* NodeType.SIZE = 9; <= 1 disables the embedding */
  int32_t node_in;            /* feature width WITHOUT the embedding: K*dim [+K] [+2dim] [+dim] */
  int32_t edge_in;            /* dim + 1 */
  int32_t out_dim;            /* particle_dimension */
} lb_gns_desc;

const char* lb_strerror(int code);
const char* lb_last_error(void);
int lb_version(void);

/* ---- engine / state ------------------------------------------------------------------ */

/* case_builder(...) - case_setup/case.py:62.  stream = hipStream_t (NULL = default stream). */
int lb_engine_create(const lb_case_desc* desc, void* hip_stream, lb_engine** out);
void lb_engine_destroy(lb_engine* eng);

/* sample[1] of the reference's (pos, particle_type) tuple: (B,N) int32. */
int lb_set_particle_type(lb_engine* eng, const int32_t* ptype_dev);

/* LB_FORCE_BUFFER only: (B,N,dim) fp64 = vmap(external_force_fn)(most_recent_position). */
int lb_set_force(lb_engine* eng, const double* force_dev);

/* Load the position window sample[0][:, t0:t0+isl] from a (B,N,T,dim) fp64 trajectory
 * (the layout H5Dataset.get_trajectory returns, data/data.py:199-225) into the engine's SoA
 * ring, and reset the step counter to `step` (normally 0).  rollout.py:113. */
int lb_load_window(lb_engine* eng, const double* traj_dev, int32_t T, int32_t t0, int32_t step);

/* Copy the current window back out as (B,N,isl,dim) fp64 (oldest frame first). */
int lb_read_window(lb_engine* eng, double* win_out_dev);

/* ---- neighbor list: jax_sph.jax_md.partition.neighbor_list (3rd party) ----------------
 * constructed case.py:120-130; used .allocate case.py:184-186, .update case.py:188-190. */

/* neighbor_fn.allocate(most_recent_position): HOST-SYNCHRONOUS sizing.  Builds the list for the
 * current window's newest frame, then freezes cell_capacity = int(max cell occupancy * mult)
 * and E_cap = int(max_b occupancy_b * mult) (clamped like jax-md).  Outputs may be NULL. */
int lb_nl_allocate(lb_engine* eng, int32_t* cell_capacity_out, int32_t* e_cap_out,
                   int32_t* occupancy_out /* [B] host */);

/* Pin the frozen capacities explicitly (e.g. to mirror a reference NeighborList). */
int lb_nl_set_capacity(lb_engine* eng, int32_t cell_capacity, int32_t e_cap);

/* neighbors.update(most_recent_position): asynchronous rebuild with the frozen capacities.
 * Sets the per-trajectory did_buffer_overflow flags (read them with lb_nl_read_flags). */
int lb_nl_update(lb_engine* eng);

/* did_buffer_overflow per trajectory -> (B,) int32 device buffer (async copy). */
int lb_nl_read_flags(lb_engine* eng, int32_t* overflow_out_dev);

/* NeighborList.idx in the reference's format: (B, 2, E_cap) int32, row 0 receivers, row 1
 * senders, trajectory-local ids, padding = N (features.py:110).  Edge ORDER is the engine's
 * canonical one - sorted by (receiver, sender) - not jax-md's slot order.  n_edges_out_dev
 * (B,) int32 gets the real edge counts (may be NULL). */
int lb_nl_read_idx(lb_engine* eng, int32_t* idx_out_dev, int32_t* n_edges_out_dev);

/* ---- features: feature_transform - case_setup/features.py:47-126 ---------------------- */

/* Node features as fp64 in reference column order:
 *   vel_hist (B,N,K*dim) time-major; vel_mag (B,N,K) or NULL; bound (B,N,2*dim) or NULL;
 *   force (B,N,dim) or NULL. */
int lb_node_features(lb_engine* eng, double* vel_hist_out_dev, double* vel_mag_out_dev,
                     double* bound_out_dev, double* force_out_dev);

/* Edge features of the CURRENT list in lb_nl_read_idx order: rel_disp (B,E_cap,dim) and
 * rel_dist (B,E_cap,1) fp64.  Padded rows hold what the reference's clamped gather produces
 * (disp(pos[N-1], pos[N-1]) = 0). */
int lb_edge_features(lb_engine* eng, double* rel_disp_out_dev, double* rel_dist_out_dev);

/* ---- model: GNS - models/gns.py:35-171 ------------------------------------------------ */

/* Weights are fp32 HOST arrays in haiku layout (Linear w is (in,out) row-major), concatenated in
 * module creation order:
 *   [embed (types,emb)] then for each MLP in order enc_node, enc_edge, (proc_k_edge, proc_k_node)
 *   for k < num_mp_steps, decoder:  w0, b0, w1, b1, [ln_scale, ln_offset]  (decoder: no LN).
 * n_floats is the total length (checked). */
int lb_gns_create(lb_engine* eng, const lb_gns_desc* desc, const float* weights_host,
                  int64_t n_floats, lb_gns** out);
void lb_gns_destroy(lb_gns* gns);

/* model.apply(params, state, (features, particle_type)) -> {"acc": (B,N,dim) fp32}
 * (rollout.py:59) on the engine's current window and neighbor list. */
int lb_gns_forward(lb_engine* eng, lb_gns* gns, float* acc_out_dev);

/* 1 (default): jraph.segment_sum is fused into the edge-MLP epilogue (segmented wavefront scan,
 * no message round trip through HBM); 0: messages are written out and reduced by the stand-alone
 * segment_sum kernel (lb_segment_sum's kernel).  Both are deterministic and atomic-free. */
int lb_set_fused_aggregation(lb_engine* eng, int32_t on);

/* Debug/parity taps: node latents after the encoder and after each MP step
 * ((num_mp_steps+1), B*N, latent) fp32, or NULL. */
int lb_gns_set_tap(lb_gns* gns, float* node_latents_out_dev);

/* ---- integrator + driver -------------------------------------------------------------- */

/* _forward_eval minus the model call - rollout.py:61-73 with case.integrate (case.py:230-259):
 * next = shift(p[-1], disp(p[-1], p[-2]) + acc_mean + acc*acc_std); kinematic particles
 * (utils.py:28-35) take target (B,N,dim) fp64 instead; the window advances by one frame and
 * the step counter increments.  If pred_out_dev != NULL the new frame is also written to
 * pred_out_dev[b][step][:][:] of a (B,pred_T,N,dim) buffer (rollout.py:165-167). */
int lb_integrate(lb_engine* eng, const float* acc_dev, const double* target_dev,
                 double* pred_out_dev, int32_t pred_T);

/* case.integrate(normalized_in, position_sequence) alone - case.py:230-259 - stateless:
 * mode 0 "acc": shift(p[-1], disp(p[-1],p[-2]) + acc_mean + pred*acc_std);
 * mode 1 "vel": shift(p[-1], vel_mean + pred*vel_std).  ("pos" is the identity: host side.)
 * pred (B,N,dim) fp32; pos_seq (B,N,T,dim) fp64 (last two frames used); out (B,N,dim) fp64. */
int lb_case_integrate(lb_engine* eng, int32_t mode, const float* pred_dev, const double* pos_seq_dev,
                      int32_t T, double* next_out_dev);

/* _eval_batched_rollout's step loop - rollout.py:125-169 - fully on the device:
 * for step in [0, n_steps): nl_update -> features -> GNS -> integrate -> store prediction.
 * traj_dev (B,N,T,dim) fp64 supplies the initial window (frames [0,isl)) and the kinematic
 * targets (frame isl+step, clamped to T-1 as JAX clamps the gather, rollout.py:159).
 * HOST-SYNCHRONOUS at the end.  On neighbor overflow it re-allocates (allocate_eval semantics,
 * rollout.py:139-151) and resumes from the overflowed step; *n_realloc_out counts that. */
int lb_rollout(lb_engine* eng, lb_gns* gns, const double* traj_dev, int32_t T, int32_t n_steps,
               double* pred_out_dev, int32_t* n_realloc_out);

/* ---- SEGNN (models/segnn.py:403-610) ---------------------------------------------------------- */

/* ---- training step ----------------------------------------------------
 * trainer.py:35-89: loss = _mse (weighted squared error of the normalised accelerations, summed over dim, masked
 * to the non-kinematic particles, divided by their number), value_and_grad vmapped over the batch with the
 * gradients SUMMED and the loss averaged, optax.adamw.  Forward (with saved activations), loss, backward and the
 * optimiser step run on the device (csrc/lb_train.hip: rocBLAS sgemm for the dense contractions, hand-written HIP
 * kernels for everything else); the graph, features and targets are the engine's (case.preprocess first). */
typedef struct lb_gns_train lb_gns_train;
/* weights_host: the flat blob of GNS.flatten (same layout as lb_gns_create); latent <= 128 (narrower: zero-padded on the device),
 * two to eight Linears per MLP (num_mlp_layers, models/utils.py:100-115; round 6). */
int lb_gns_train_create(lb_engine* eng, const lb_gns_desc* desc, const float* weights_host, int64_t n_floats,
                        lb_gns_train** out);
void lb_gns_train_destroy(lb_gns_train* t);
/* jax.value_and_grad(_mse) summed over the batch, on the engine's current window + neighbor list.
 * target_dev (B*N, dim) fp32 normalised accelerations; gradients ACCUMULATE (lb_gns_train_zero_grad);
 * *loss_out = mean of the per-trajectory losses; pred_out_dev (optional) receives the (B*N, dim) predictions. */
int lb_gns_train_loss_grad(lb_gns_train* t, const float* target_dev, float loss_weight, double* loss_out,
                           float* pred_out_dev);
int lb_gns_train_zero_grad(lb_gns_train* t);
/* optax.adamw(learning_rate, b1, b2, eps, weight_decay) on every parameter (trainer.py:183-193). */
int lb_adamw_step(lb_gns_train* t, float lr, float b1, float b2, float eps, float weight_decay);
/* which: 0 weights, 1 gradients, 2 / 3 first / second AdamW moment; flat blob in GNS.flatten order.
 * write: step >= 0 also restores the optimiser's step counter (bias correction). */
int lb_gns_train_read(lb_gns_train* t, int32_t which, float* out_host, int64_t n_floats);
int lb_gns_train_write(lb_gns_train* t, int32_t which, const float* in_host, int64_t n_floats, int64_t step);
/* optax's `count`: AdamW steps taken so far (the bias-correction exponent of the NEXT step is count + 1); what a
 * checkpoint has to store to resume bit-identically (utils.py:61-91 pickles it inside opt_state). */
int64_t lb_gns_train_step_count(lb_gns_train* t);
/* Arithmetic of the training step (round 5 / 6).  Default: every tall-skinny product (Y = XW, dX = dY W^T, dW += X^T dY) runs
 * as three fp16 MFMA passes over hi / lo splits of fp32 operands, fp32 accumulate; the operands of the first two are put into
 * fp16's range by exact power-of-two scales per row block / matrix, dY of the third per row chunk.  Its X operand (saved
 * activations) carries ONE power-of-two scale per call site (initially 1) under a range GUARD: when a chunk's largest scaled
 * |X| leaves [2^-8, 2^15) the step's gradients are NOT added, lb_gns_train_loss_grad re-centres the scales of the call sites
 * that fired and repeats the step (a call site that fires again switches to per-chunk scales: one more pass over its X).
 * Error per product term <= 2^-22 of the operand chunks' scales; gradients vs float64 autograd <= 1e-4 per leaf
 * (tests/test_train.py).  LB_TRAIN_MATH=f32 in the environment at handle creation selects the exact-fp32 MFMA kernels
 * throughout (1.7x slower).  This returns how many steps were repeated so far. */
int32_t lb_gns_train_math_fallbacks(lb_gns_train* t);

typedef struct lb_segnn lb_segnn;

/* SEGNN hyper-parameters (runner.py:217-237; configs: scalar_units 64 -> hidden irreps
 * 32x0e+32x1o through weight_balanced_irreps, segnn.py:365-400). */
typedef struct lb_segnn_desc {
  int32_t hidden;           /* multiplicity n of every irrep of the hidden irreps n x (0e + 1o + ..) (weight_balanced_irreps) */
  int32_t blocks_per_step;  /* num_mlp_layers (2) */
  int32_t num_mp_steps;
  int32_t homogeneous;      /* homogeneous_particles: 1 = no one-hot particle-type scalars */
  int32_t n_vels;           /* input_seq_length - 1 */
  int32_t velocity_avg;     /* velocity_aggregate: 1 "avg", 0 "last" */
  int32_t lmax_hidden;      /* 0 .. 2 (segnn.py:482-484) */
  int32_t lmax_attributes;  /* 0 .. 2: attributes = spherical harmonics up to this order (segnn.py:481) */
  int32_t norm;             /* segnn_norm: 0 None, 1 "instance", 2 "batch" (segnn.py:303,346-351) */
  float norm_eps;           /* eps of e3nn's BatchNorm; <= 0: 1e-5 */
} lb_segnn_desc;

/* SEGNN(...) + params.  Two weight layouts.
 * (1) hidden == 32, lmax_hidden == lmax_attributes == 1, norm == 0 (every shipped config; the fused kernels):
 *   weights_host: one (ws, wv, b) triple per O3TensorProduct in call order
 * (embedding_nodes; per layer message tp_0.. then update tp_0..; readout_0..; output):
 *   ws (K, Ms) weights of the 0e outputs (gated blocks: Ms = 2*hidden, activated scalars first,
 *   then the gates), wv (K, Mv) weights of the 1o outputs, b (Ms).  K indexes the tensor-product
 *   channels operand by operand, scalar-derived channels first, then vector-derived
 *   (oracle/segnn_oracle.py:tp_inputs); the 1/sqrt(K) of e3nn's Linear is applied here.
 * (2) anything else with lmax <= 2 (csrc/lb_segnn_gen.hip): per O3TensorProduct in the same call order, per output irrep
 *   l = 0, 1, 2: W_l (K_l, mul_l) with the rows in e3nn's regrouped order (x chunk major, attribute l minor), then b (the
 *   scalar outputs) if the output has 0e; then, with norm, per layer [messages: weight, bias - "batch" only], nodes:
 *   weight (one per channel of every hidden irrep), bias (one per hidden scalar) (e3nn BatchNorm, training-mode statistics
 *   as the reference calls it). */
int lb_segnn_create(lb_engine* eng, const lb_segnn_desc* desc, const float* weights_host,
                    int64_t n_floats, lb_segnn** out);

/* The same training step for SEGNN (round 5; reference: train/trainer.py:35-89 is model-agnostic, models/segnn.py:44-362,
 * 595-610 is the network).  The handle type is lb_gns_train (round 3's name for "a training handle"): lb_gns_train_zero_grad,
 * lb_adamw_step, lb_gns_train_read / _write (flat blob in SEGNN.flatten order = lb_segnn_create's), lb_gns_train_step_count
 * and lb_gns_train_destroy work on it unchanged; lb_gns_train_loss_grad dispatches to lb_segnn_train_loss_grad.
 * hidden <= 32, lmax 1, norm None (the shipped configs); gradients bit-reproducible. */
int lb_segnn_train_create(lb_engine* eng, const lb_segnn_desc* desc, const float* weights_host, int64_t n_floats,
                          lb_gns_train** out);
int lb_segnn_train_loss_grad(lb_gns_train* t, const float* target_dev, float loss_weight, double* loss_out,
                             float* pred_out_dev);
void lb_segnn_destroy(lb_segnn* segnn);

/* SEGNN.__call__ -> {"acc": (B,N,dim) fp32} (segnn.py:595-610) on the current window + list. */
int lb_segnn_forward(lb_engine* eng, lb_segnn* segnn, float* acc_out_dev);

/* Arithmetic of the GNS GEMMs.  Default (LB_MATH unset) = mode 1: every fp32 operand is carried as an fp16
 * hi/lo pair on the fp16 MFMA (fp32-class accuracy, ~5x fewer matrix-pipe cycles than the fp32 MFMA) WITH a
 * range guard: operands >= 2^15 (sampled), operand ROWS whose values all sit below 2^-11 (tested on every tile of the
 * batch edge / node kernels and of the small-graph kernels), and non-finite accelerations raise
 * flags; lb_gns_forward / lb_rollout then redo the flagged forward / rollout step in mode 0 and return to mode 1
 * (round 4: not sticky; after LB_GUARD_MAX_FALLBACKS = 3 flagged steps the rest of that rollout runs in mode 0).
 * set_mode: -1 query, 0 exact fp32 MFMA, 1 guarded f16x2, 2 f16x2 without the switch (tests), 3 guarded with
 * every k-group of every tile tested (LB_GUARD=full).  flags_out: guard bits raised and not yet consumed
 * (1 large, 2 tiny, 4 non-finite).  The reference computes the model in fp32 (runner.py:71-72). */
int lb_math_mode(lb_engine* eng, int32_t set_mode, int32_t* mode_out, int32_t* flags_out);

/* Steps (or stand-alone forwards) the range guard has redone in exact fp32 since the engine was created. */
int32_t lb_math_fallbacks(lb_engine* eng);

/* What the per-tile guard covers (mode 1): LARGE / NaN - the sampled probe (first tile of every wave) on every GEMM operand,
 * plus the decoder's output test on every row; TINY rows - the edge latents and the hidden layer of k_edge16v, every operand
 * of the M-split kernels, the hidden layer of k_node16s.  NOT row-tested: the first operand [node latents | aggregated
 * messages] of k_node16s (its rows are LayerNorm outputs, |x| = O(1), plus sums of them).  A flagged step is redone in fp32
 * and the steps before it are kept; with LB_GUARD=sampled (probe only) the rollout restarts from step 0 instead.
 *
 * Test hook: the NEXT lb_rollout behaves as if the guard had raised `flags` (LB_MATH_* bits: 1 large, 2 tiny, 4 non-finite)
 * at rollout step `step` (one-shot; step < 0 disarms).  Lets a test drive the redo-one-step-in-fp32 / resume-in-f16x2
 * path on healthy weights and compare the result with the oracle.  No reference counterpart. */
int lb_debug_inject_guard(lb_engine* eng, int32_t flags, int32_t step);

/* Debug/parity tap: hidden node state after the embedding and after each layer,
 * ((num_mp_steps+1), B*N, lb_segnn_row_floats) fp32 rows, or NULL.  Layout (1): 128 floats [s(32) | vx(32) | vy(32) | vz(32)];
 * layout (2): e3nn's own layout (n scalars, n x 3, n x 5), padded to a multiple of 4 floats. */
int lb_segnn_set_tap(lb_segnn* segnn, float* hidden_out_dev);
int32_t lb_segnn_row_floats(lb_segnn* segnn);

/* lb_rollout with SEGNN as the model. */
int lb_segnn_rollout(lb_engine* eng, lb_segnn* segnn, const double* traj_dev, int32_t T,
                     int32_t n_steps, double* pred_out_dev, int32_t* n_realloc_out);

/* MetricsComputer.mse / .mae per step - evaluate/metrics.py:139-147: mean over (N,dim) of
 * disp(pred,target)^2 (|.|) with the case's displacement.  Both rollouts are (B,T,N,dim) fp64,
 * the layout metrics_computer receives (rollout.py:171-176).  Outputs (B,n_steps) fp64,
 * either may be NULL. */
int lb_metrics(lb_engine* eng, const double* pred_dev, int32_t pred_T, const double* target_dev,
               int32_t target_T, int32_t n_steps, double* mse_out_dev, double* mae_out_dev);

/* MetricsComputer "e_kin" - evaluate/metrics.py:98-125,157-160: kinetic energy of strided frames of
 * one rollout (B,T,N,dim) fp64: out[b][k] = dx^dim * sum((disp(x[1+k*stride], x[k*stride]) / dt)^2),
 * k < n_out = ceil((T-1)/stride) (= len(x[1::stride])).  out (B,n_out) fp64. */
int lb_ekin(lb_engine* eng, const double* rollout_dev, int32_t T, int32_t stride, double dt, double dx,
            double* out_dev, int32_t n_out);

/* MetricsComputer "sinkhorn" - evaluate/metrics.py:127-136,162-176,198-213: Sinkhorn divergence
 * between the predicted and the target particle cloud of every stride-th frame,
 * out[b][k] = S_eps(pred[b, k*stride], target[b, k*stride]), k < n_out = ceil(T/stride) with
 * T = min(pred_T, target_T) frames compared.  Cost = squared case displacement rounded to float32,
 * uniform weights, eps = 0.05 * mean(C_xy) shared by the xy / xx / yy problems, convergence
 * `threshold` (the reference passes 1e-4) on the L1 marginal error checked every 10 iterations
 * (ott-jax 0.4.6 sinkhorn_divergence, restated in oracle/sinkhorn_oracle.py).  Rollouts (B,T,N,dim)
 * fp64; out (B,n_out) fp64 on the device; iters_out_host (optional, HOST, B*n_out*3 int32) receives
 * the iteration counts of the three solves.  Host-synchronous (the stopping rule is data dependent). */
int lb_sinkhorn(lb_engine* eng, const double* pred_dev, int32_t pred_T, const double* target_dev,
                int32_t target_T, int32_t stride, double threshold, double* out_dev, int32_t n_out,
                int32_t* iters_out_host);

/* MetricsComputer(ot_backend="pot") - evaluate/metrics.py:178-196: POT's sinkhorn2(a, b, M, reg, numItermax,
 * stopThr) (the reference passes reg=0.1, numItermax=500, stopThr=1e-05) on the xy / xx / yy float32 cost
 * matrices with uniform weights, out[b][k] = clip(xy - 0.5 * (xx + yy), 0) evaluated in float32 like the reference
 * (each sinkhorn2 value is rounded to float32 first).  POT is an optional dependency of the reference; its
 * Sinkhorn-Knopp iteration is restated in oracle/sinkhorn_pot_oracle.py.  Same frame selection / shapes as
 * lb_sinkhorn.  info_out_host (optional, HOST, B*n_out*6 int32): iterations of the three solves, then how each
 * loop ended (0 numItermax reached, 1 err < stopThr, 2 numerical stop: previous scalings kept). */
int lb_sinkhorn_pot(lb_engine* eng, const double* pred_dev, int32_t pred_T, const double* target_dev,
                    int32_t target_T, int32_t stride, double reg, int32_t num_iter_max, double stop_thr,
                    double* out_dev, int32_t n_out, int32_t* info_out_host);

/* ---- measurement hooks (bench.py) ------------------------------------------------------ */

/* Enable per-kernel-class HIP-event timing on the engine stream.  Classes: see lb_timer_name. */
int lb_timers_enable(lb_engine* eng, int32_t on);
int lb_timers_reset(lb_engine* eng);
int32_t lb_timer_count(void);
const char* lb_timer_name(int32_t cls);
/* Accumulated milliseconds and launch count of one class (host-synchronous). */
int lb_timer_get(lb_engine* eng, int32_t cls, double* ms_out, int64_t* launches_out);

/* Current totals: real edges over all trajectories, E_cap, cell capacity (host-synchronous). */
int lb_stats(lb_engine* eng, int64_t* n_edges_total, int32_t* e_cap, int32_t* cell_capacity);

/* Edge accounting for measurements: every neighbor-list build (allocate or update, each rollout step is one) adds its real
 * edge count (all B trajectories, before clamping to the capacity) to a device-side sum.  sum_edges / n_builds = the MEAN
 * E of the steps run since the last reset - what a per-launch byte or flop count averaged over a rollout must use
 * (SURVEY 8d "use real E"; a rollout's E drifts).  first_edges / last_edges: the first build after the reset and the
 * latest one.  reset != 0 zeroes the sums after reading.  Host-synchronous.  No reference counterpart. */
int lb_edge_accounting(lb_engine* eng, int64_t* sum_edges, int64_t* n_builds, int64_t* first_edges, int64_t* last_edges,
                       int32_t reset);

/* Which kernels a GNS forward of this engine runs on at its CURRENT size / arithmetic mode, as
 * "edge=<kernel>;node=<kernel>" (NUL-terminated, truncated to cap): the network kernels are picked by graph size
 * (M-split kernels for one small trajectory, wave-per-tile kernels for batches) - bench.py names the kernel its
 * roofline line is about with this.  No reference counterpart (measurement aid). */
int lb_kernel_names(lb_engine* eng, char* out, int32_t cap);

/* Stand-alone jraph.segment_sum(messages, receivers, N) on the CURRENT list (gns.py:117-119):
 * msg (E_total,D) fp32 in engine edge order -> out (B*N,D) fp32.  Used by tests and by the
 * aggregation-roofline leg of bench.py. */
int lb_segment_sum(lb_engine* eng, const float* msg_dev, float* out_dev, int32_t D);

#ifdef __cplusplus
}
#endif
#endif /* LBHIP_H */
