// lb_device.h - small device helpers shared by the kernels.  The fp64 geometry follows the CPU
// oracle operation-for-operation (library built with -ffp-contract=off) so that the neighbor
// predicate, features and integrator are bit-identical to it.
#pragma once
#include "lb_internal.h"

// dtype=float32 geometry (case.py:169 casts the positions to the caller's dtype; the reference's own
// tests/case_test.py runs in f32).  Buffers stay fp64; in f32 mode every stored value is float-representable and every
// arithmetic result is rounded to float: for +, -, *, / and sqrt a double operation on float operands followed by
// one rounding to float IS the correctly rounded float operation (53 >= 2*24 + 2 bits), so the results are those
// of a float pipeline bit for bit.  f32 is a compile-time constant in the neighbor search, a run-time flag elsewhere.
__device__ __forceinline__ double lb_r(double x, int f32) { return f32 ? (double)(float)x : x; }

// jnp.mod(x, L) for L > 0: C fmod, then shift a negative remainder by +L
// (jax_md.space.periodic_displacement / periodic_shift use it).
__device__ __forceinline__ double lb_mod(double x, double L, int f32 = 0) {
  double ax = fabs(x);
  double r;
  if (ax < L)
    r = x;  // fmod(x, L) == x
  else if (ax < 2.0 * L)
    r = (x > 0.0) ? (x - L) : (x + L);  // exact (Sterbenz), in either precision
  else
    r = fmod(x, L);
  if (r != 0.0 && r < 0.0) r = lb_r(r + L, f32);
  return r;
}

// displacement_fn(a, b) of case.py:104-108: periodic -> mod(a-b+L/2, L) - L/2, free -> a-b.
__device__ __forceinline__ double lb_disp1(double a, double b, double L, double halfL,
                                           int periodic, int f32 = 0) {
  double d = lb_r(a - b, f32);
  if (!periodic) return d;
  return lb_r(lb_mod(lb_r(d + halfL, f32), L, f32) - halfL, f32);
}

// shift_fn(r, dr): periodic -> mod(r+dr, L), free -> r+dr.
__device__ __forceinline__ double lb_shift1(double r, double dr, double L, int periodic, int f32 = 0) {
  double s = lb_r(r + dr, f32);
  return periodic ? lb_mod(s, L, f32) : s;
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
