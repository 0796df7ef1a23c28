// lb_device.h - small device helpers shared by the kernels.  The fp64 geometry follows the CPU
// oracle operation-for-operation (library built with -ffp-contract=off) so that the neighbor
// predicate, features and integrator are bit-identical to it.
#pragma once
#include "lb_internal.h"

// jnp.mod(x, L) for L > 0: C fmod, then shift a negative remainder by +L
// (jax_md.space.periodic_displacement / periodic_shift use it).
__device__ __forceinline__ double lb_mod(double x, double L) {
  double ax = fabs(x);
  double r;
  if (ax < L)
    r = x;  // fmod(x, L) == x
  else if (ax < 2.0 * L)
    r = (x > 0.0) ? (x - L) : (x + L);  // exact (Sterbenz)
  else
    r = fmod(x, L);
  if (r != 0.0 && r < 0.0) r += L;
  return r;
}

// displacement_fn(a, b) of case.py:104-108: periodic -> mod(a-b+L/2, L) - L/2, free -> a-b.
__device__ __forceinline__ double lb_disp1(double a, double b, double L, double halfL,
                                           int periodic) {
  double d = a - b;
  if (!periodic) return d;
  return lb_mod(d + halfL, L) - halfL;
}

// shift_fn(r, dr): periodic -> mod(r+dr, L), free -> r+dr.
__device__ __forceinline__ double lb_shift1(double r, double dr, double L, int periodic) {
  double s = r + dr;
  return periodic ? lb_mod(s, L) : s;
}

__device__ __forceinline__ int lb_slot(int frame, int isl) { return frame % isl; }

// position component d of particle gidx in the ring frame `frame` (0 = oldest of the window
// that starts at step `step`): slot = (step + frame) % isl.
__device__ __forceinline__ double lb_pos(const double* __restrict__ win, const lb_geom& g,
                                         int64_t BN, int step, int frame, int d, int64_t gidx) {
  int slot = (step + frame) % g.isl;
  return win[((int64_t)slot * g.dim + d) * BN + gidx];
}

__device__ __forceinline__ int lb_lane() { return threadIdx.x & 63; }

// range guard: raise flag bits and remember the first rollout step they were raised in
__device__ __forceinline__ void lb_raise_math(const lb_ctrl* ctrl, int flags) {
  lb_ctrl* c = const_cast<lb_ctrl*>(ctrl);
  atomicOr(&c->math_flags, flags);
  atomicMin(&c->math_step, c->step);
}
