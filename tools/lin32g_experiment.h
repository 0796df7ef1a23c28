// tools/lin32g_experiment.h - EXPERIMENT (not in the library): the plain product of k_lin32f on v_mfma_f32_32x32x2_f32 with
// 32-row tiles - half the LDS operand traffic per flop (one ds_read_b128 feeds 4 MFMAs of 4096 flops instead of 2048).
// Question: does that raise the sustained rate of a kernel that sits at 0.5 of the 2.4 GHz fp32 MFMA peak under the power cap?
#pragma once
#include "lb_lin32.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

// fragment order for the 32 x 32 x 2 scheme: f32x4 entry ((g * 4 + cb) * 64 + lane)[e] = Wop[8 g + 4 (lane >> 5) + e][32 cb + (lane & 31)]
__global__ void __launch_bounds__(256) k_pack_w32(const float* __restrict__ w, float* __restrict__ wp, int NR, int NO, int ldw,
                                                  int trans) {
  const int NG = NR / 8, total = NG * 4 * 256;
  for (int idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
    const int e = idx & 3, ln = (idx >> 2) & 63, q = idx >> 8;
    const int cb = q & 3, g = q >> 2;
    const int k = 8 * g + 4 * (ln >> 5) + e, m = 32 * cb + (ln & 31);
    float v = 0.f;
    if (k < NR && m < NO) v = trans ? w[(int64_t)m * ldw + k] : w[(int64_t)k * ldw + m];
    wp[idx] = v;
  }
}

__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) k_lin32g(lb_lin_args a) {
  extern __shared__ f32x4 sWl[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = lane & 31, h = lane >> 5;
  const int NG = a.NR >> 3, nch = NG >> 3;   // k-groups of 8, chunks of 8 groups
  const int64_t ntiles = (a.rows + 31) >> 5, tstep = (int64_t)gridDim.x * 8;
  int64_t t = (int64_t)wave * gridDim.x + blockIdx.x;
  auto row_ptr = [&](int64_t tt) -> const float* {
    int64_t r = tt * 32 + n;
    r = r < a.rows ? r : a.rows - 1;
    return a.X + r * a.ldx + 4 * h;
  };
  const float* xr = row_ptr(t);
  f32x4 ring[8];
#pragma unroll
  for (int jj = 0; jj < 8; ++jj) ring[jj] = *reinterpret_cast<const f32x4*>(xr + 8 * jj);
  lb_lin_stage(sWl, a.Wp, NG * 4 * 64, tid);
  __syncthreads();
  const f32x4* sw0 = sWl + lane;
  f32x4 wv[2][4];
#pragma unroll
  for (int cb = 0; cb < 4; ++cb) wv[0][cb] = sw0[cb * 64];
  for (; t < ntiles; t += tstep) {
    const int64_t row = t * 32 + n;
    const bool live = row < a.rows;
    const float* xnext = row_ptr(t + tstep);
    f32x16 acc[4];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[cb][v] = 0.f;
    for (int jc = 0; jc < nch; ++jc) {
      const bool more = jc + 1 < nch;
      const float* nx = more ? xr + 64 * (jc + 1) : xnext;
      const f32x4* sw = sw0 + (jc * 8 * 4) * 64;
      const f32x4* swn = more ? sw + 8 * 4 * 64 : sw0;
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) {
        const f32x4* sn = jj < 7 ? sw + (jj + 1) * 4 * 64 : swn;
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) wv[(jj + 1) & 1][cb] = sn[cb * 64];
        __builtin_amdgcn_sched_barrier(0);
        const f32x4 x = ring[jj];
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int cb = 0; cb < 4; ++cb)
            acc[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[jj & 1][cb][e], x[e], acc[cb], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        ring[jj] = *reinterpret_cast<const f32x4*>(nx + 8 * jj);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (live) {
      // D layout: lane (n = data row, h): register v holds output column 32 cb + 8 (v / 4) + 4 h + (v % 4)
      float* yr = a.Y + row * a.ldy + 4 * h;
#pragma unroll
      for (int cb = 0; cb < 4; ++cb)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f32x4 y = {acc[cb][4 * q], acc[cb][4 * q + 1], acc[cb][4 * q + 2], acc[cb][4 * q + 3]};
          if (a.bias) y = y + *reinterpret_cast<const f32x4*>(a.bias + 32 * cb + 8 * q + 4 * h);
          if (a.relu) {
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) y[jj] = fmaxf(y[jj], 0.f);
          }
          *reinterpret_cast<f32x4*>(yr + 32 * cb + 8 * q) = y;
        }
    }
    xr = xnext;
  }
}
