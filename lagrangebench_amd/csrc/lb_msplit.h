// lb_msplit.h - argument blocks and launchers of the M-split small-graph kernels (lb_msplit.hip).
#pragma once
#include "lb_internal.h"

struct lb_ems_args {      // edge kernels
  const lb_ctrl* ctrl;
  const int32_t* senders;
  const int32_t* receivers;
  const float* efeat;     // encoder input [E][8]
  float* elat;            // tile-blocked edge latents, in place
  const float* psr;       // [BN][256] = [n@Ws | n@Wr + b0]
  const float* w;         // lb_pack_ms images: W0 (processor: edge rows of W0, 4 k-blocks; encoder: 1) then W1
  const float* b0;        // encoder only
  const float* b1;
  const float* ln_s;
  const float* ln_o;
  float* agg;             // [BN][128] rows complete inside one tile
  float* part;            // [tiles][2][128] segments cut by a tile boundary
  int skip_elat_store;    // last processor layer: the updated edge latents have no reader
  long long* dbg;         // -DLB_MS_STAMPS builds: shader-clock stamps of workgroup 0 (null in the product path)
};

struct lb_nms_args {      // node kernel
  const lb_ctrl* ctrl;
  int64_t n_rows;
  const float* xin;       // [rows][32*NKA]
  const float* agg;       // [rows][128]
  const int32_t* row_ptr;
  const float* part;
  int fused;              // agg comes from the fused edge epilogue (agg + per-tile partial slots)
  float* nlat;            // out [rows][128]
  const float* w;         // lb_pack_ms images: W0 (NKA [+4] k-blocks), W1, [projection 128 x 256]
  const float* b0;
  const float* b1;
  const float* ln_s;
  const float* ln_o;
  const float* bp;        // [256] projection bias
  float* psr;             // out [rows][256]
  // DEC (last processor layer): the decoder MLP (gns.py:125-133) and, in a rollout step, the integrator run in this
  // launch's epilogue; `w` then continues with [decoder W0 | out_dim block of decoder W1 (scaled, see dec_unscale)]
  const float* bd0;       // [128] decoder hidden bias
  const float* bd1;       // [>= 4] decoder output bias
  float dec_unscale;
  float* acc_out;         // [rows][4]
  int32_t out_dim;
  lb_integ_job integ;     // on = 0: stand-alone forward
  long long* dbg;         // -DLB_MS_STAMPS builds: shader-clock stamps of workgroup 0 (null in the product path)
};

void lb_pack_ms(const float* w, int K, int M, int nkb, int npw, bool perm, float* out);
int lbk_edge_ms(lb_engine* e, const lb_ems_args& a);
int lbk_edge_enc_ms(lb_engine* e, const lb_ems_args& a);
int lbk_node_ms(lb_engine* e, const lb_nms_args& a, int nka, bool agg, bool resid, bool proj, bool dec = false);
