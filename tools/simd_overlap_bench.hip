// tools/simd_overlap_bench.hip - can a SIMD's matrix pipe and its VALU run at the same time when the
// work comes from DIFFERENT waves of that SIMD?  (The edge kernel's counters read MFMA busy 44 % +
// VALU busy 48 % of the kernel time - is that a sum that cannot overlap, or lost overlap?)
// One 512-thread workgroup per CU = two waves per SIMD (waves w and w+4 share a SIMD: 8 waves are
// dealt round-robin to 4 SIMDs).  Roles: M = 16x16x32 f16 MFMA chain on 4 accumulators,
// V = plain v_fma_f32 on 16 registers, P = v_pk_fma_f32, D = v_fmac_f32_dpp row_shr, L = ds_read_b128.
// Prints cycles per instruction for each role alone and for pairs sharing every SIMD.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/simd_overlap_bench.hip -o tools/bin/simd_overlap_bench
#include <hip/hip_runtime.h>

#include <cstdio>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

enum { R_NONE = 0, R_MFMA, R_VALU, R_PK, R_DPP, R_LDS, R_MFMA_VALU };

template <int ROLE>
__device__ __forceinline__ float role_body(int iters, float seed, const f32x4* lds) {
  float out = 0.f;
  if (ROLE == R_MFMA || ROLE == R_MFMA_VALU) {
    f32x4 acc[4] = {{seed, 0, 0, 0}, {0, seed, 0, 0}, {0, 0, seed, 0}, {0, 0, 0, seed}};
    h8 a, b;
    for (int i = 0; i < 8; ++i) {
      a[i] = (_Float16)(seed + i);
      b[i] = (_Float16)(seed - i);
    }
    float v[8] = {seed, 1, 2, 3, 4, 5, 6, 7};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[c], 0, 0, 0);
        if (ROLE == R_MFMA_VALU) {  // 8 independent fmas per 4 MFMAs in the SAME wave
#pragma unroll
          for (int k = 0; k < 8; ++k) v[k] = __builtin_fmaf(v[k], 1.0001f, 0.5f);
        }
      }
    }
    for (int c = 0; c < 4; ++c) out += acc[c][0] + acc[c][3];
    for (int k = 0; k < 8; ++k) out += v[k];
  } else if (ROLE == R_VALU) {
    float v[16];
    for (int k = 0; k < 16; ++k) v[k] = seed + k;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = __builtin_fmaf(v[k], 1.0001f, 0.5f);
    }
    for (int k = 0; k < 16; ++k) out += v[k];
  } else if (ROLE == R_PK) {
    f32x2 v[16];
    for (int k = 0; k < 16; ++k) v[k] = f32x2{seed + k, seed - k};
    const f32x2 m = {1.0001f, 0.9999f}, c = {0.5f, 0.25f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = __builtin_elementwise_fma(v[k], m, c);
    }
    for (int k = 0; k < 16; ++k) out += v[k][0] + v[k][1];
  } else if (ROLE == R_DPP) {
    float v[16];
    for (int k = 0; k < 16; ++k) v[k] = seed + k;
    const float m = 0.5f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        asm volatile(
            "v_fmac_f32_dpp %0, %0, %16 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
            "v_fmac_f32_dpp %1, %1, %16 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
            "v_fmac_f32_dpp %2, %2, %16 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
            "v_fmac_f32_dpp %3, %3, %16 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
            "v_fmac_f32_dpp %4, %4, %16 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
            "v_fmac_f32_dpp %5, %5, %16 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
            "v_fmac_f32_dpp %6, %6, %16 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
            "v_fmac_f32_dpp %7, %7, %16 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
            "v_fmac_f32_dpp %8, %8, %16 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
            "v_fmac_f32_dpp %9, %9, %16 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
            "v_fmac_f32_dpp %10, %10, %16 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
            "v_fmac_f32_dpp %11, %11, %16 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
            "v_fmac_f32_dpp %12, %12, %16 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
            "v_fmac_f32_dpp %13, %13, %16 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
            "v_fmac_f32_dpp %14, %14, %16 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
            "v_fmac_f32_dpp %15, %15, %16 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
            : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]),
              "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15])
            : "v"(m));
      }
    }
    for (int k = 0; k < 16; ++k) out += v[k];
  } else if (ROLE == R_LDS) {
    f32x4 s = {0, 0, 0, 0};
    const f32x4* p = lds + (threadIdx.x & 63);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const f32x4 t = p[64 * k];
        asm volatile("" ::"v"(t));
        s[0] += 0.f;
      }
    }
    out = s[0];
  }
  return out;
}

template <int ROLE_A, int ROLE_B>
__global__ void __launch_bounds__(512, 2) k_pair(int iters, float seed, float* out, long long* cyc) {
  __shared__ f32x4 lds[1024];
  for (int i = threadIdx.x; i < 1024; i += 512) lds[i] = f32x4{seed, 1, 2, 3};
  __syncthreads();
  const int wave = threadIdx.x >> 6;
  const long long t0 = __builtin_readcyclecounter();
  float r;
  if (wave < 4)
    r = role_body<ROLE_A>(iters, seed, lds);
  else
    r = role_body<ROLE_B>(iters, seed, lds);
  const long long t1 = __builtin_readcyclecounter();
  if (r == 12345.678f) out[0] = r;
  if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) cyc[wave] = t1 - t0;
}

template <int A, int B>
static void run(const char* name, int n_a, int n_b) {
  const int iters = 2000;
  float* out;
  long long* cyc;
  (void)hipMalloc(&out, 64);
  (void)hipMalloc(&cyc, 64);
  hipLaunchKernelGGL((k_pair<A, B>), dim3(256), dim3(512), 0, 0, iters, 1.0f, out, cyc);
  hipLaunchKernelGGL((k_pair<A, B>), dim3(256), dim3(512), 0, 0, iters, 1.0f, out, cyc);
  (void)hipDeviceSynchronize();
  long long h[8];
  (void)hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
  printf("%-46s", name);
  if (n_a) printf("  A: %6.2f cyc/inst", (double)h[0] / ((double)iters * n_a));
  if (n_b) printf("  B: %6.2f cyc/inst", (double)h[4] / ((double)iters * n_b));
  printf("\n");
  (void)hipFree(out);
  (void)hipFree(cyc);
}

int main() {
  // s_memtime / readcyclecounter ticks are shader cycles
  run<R_MFMA, R_NONE>("MFMA alone (1 wave/SIMD)", 16, 0);
  run<R_MFMA, R_MFMA>("MFMA + MFMA (2 waves/SIMD)", 16, 16);
  run<R_VALU, R_NONE>("v_fma_f32 alone", 64, 0);
  run<R_VALU, R_VALU>("v_fma_f32 + v_fma_f32", 64, 64);
  run<R_PK, R_NONE>("v_pk_fma_f32 alone", 64, 0);
  run<R_DPP, R_NONE>("v_fmac_f32_dpp alone", 64, 0);
  run<R_LDS, R_NONE>("ds_read_b128 alone", 16, 0);
  run<R_MFMA, R_VALU>("MFMA (A) beside v_fma_f32 (B), same SIMD", 16, 64);
  run<R_MFMA, R_PK>("MFMA (A) beside v_pk_fma_f32 (B)", 16, 64);
  run<R_MFMA, R_DPP>("MFMA (A) beside v_fmac_f32_dpp (B)", 16, 64);
  run<R_MFMA, R_LDS>("MFMA (A) beside ds_read_b128 (B)", 16, 16);
  run<R_MFMA_VALU, R_NONE>("MFMA + 2 v_fma per MFMA in ONE wave (per MFMA)", 16, 0);
  run<R_MFMA_VALU, R_MFMA_VALU>("the same, 2 waves/SIMD (per MFMA)", 16, 16);
  return 0;
}
