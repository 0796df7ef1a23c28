// tools/mfma_chain_repro.hip - reproducer for the "MFMA accumulate-chain hazard" of profiles/HISTORY.md section 4.
//
// Round 1 observed (building k_sg_msg): a chain acc = v_mfma_f32_16x16x32_f16(a, b, acc) whose dependent
// links are separated by only ~2 independent MFMAs intermittently LOSES a link's contribution when the
// compiler rotates the accumulator registers (vDst != SrcC).  This program isolates the variables:
//   GAP   independent MFMAs between two dependent links (0..3; other accumulators)
//   ROT   0: in-place accumulation (vDst == SrcC, inline asm);  2: the plain intrinsic (compiler's choice);
//         k_partial: vDst overlapping SrcC by two registers (the "rotated" allocation), fixed registers
//   NOP   s_nop states inserted between dependent links by hand (0 or 16)
// Every lane's chain has a closed-form result (all-ones operands: each link adds K = 32 to every
// accumulator entry), so a lost link is a result that is short by a multiple of 32.  All 256 CUs run 8
// waves each, the waves of a SIMD compete for the matrix pipe (the arbitration jitter the original
// failure needed).  Prints the number of wrong accumulator entries per variant.
// RESULT (MI355X, ROCm 7.2, profiles/r02_mfma_chain_repro.txt): zero wrong entries in every variant - back to
// back, in place, partially overlapping, compiler-allocated.  The hazard is NOT reproducible in isolation;
// what the inline-asm variants DO show when the hand-inserted s_nop after the VALU zero-initialisation is
// removed is the documented VALU-write -> MFMA-SrcC-read hazard (software wait states), which is the likely
// cause of the round-1 observation in a kernel that mixed inline asm with MFMAs.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/mfma_chain_repro.hip -o tools/bin/mfma_chain_repro
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

#define LINKS 64

template <int GAP, int ROT, int NOP>
__global__ void __launch_bounds__(512) k_chain(int reps, unsigned long long* bad) {
  h8 a, b;
  for (int i = 0; i < 8; ++i) {
    a[i] = (_Float16)1.0f;
    b[i] = (_Float16)1.0f;
  }
  unsigned long long wrong = 0;
  for (int r = 0; r < reps; ++r) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f}, alt = {0.f, 0.f, 0.f, 0.f};
    f32x4 side[3] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    // a little lane/iteration dependent skew so that waves do not march in lock step
    for (int s = 0; s < ((threadIdx.x >> 6) + r) % 5; ++s) asm volatile("s_nop 7");
    // the zero-initialisation above is VALU: inline-asm MFMAs are invisible to the hazard recogniser, so
    // the VALU-write -> MFMA-SrcC-read wait states are inserted by hand
    asm volatile("s_nop 7" : "+v"(acc), "+v"(alt), "+v"(side[0]), "+v"(side[1]), "+v"(side[2]));
#pragma unroll
    for (int l = 0; l < LINKS; ++l) {
      if (ROT == 0) {
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
      } else if (ROT == 1) {
        if (l & 1)
          asm volatile("v_mfma_f32_16x16x32_f16 %0, %2, %3, %1" : "=&v"(acc) : "v"(alt), "v"(a), "v"(b));
        else
          asm volatile("v_mfma_f32_16x16x32_f16 %0, %2, %3, %1" : "=&v"(alt) : "v"(acc), "v"(a), "v"(b));
      } else {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
      }
#pragma unroll
      for (int g = 0; g < GAP; ++g) {
        if (ROT == 2)
          side[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, side[g], 0, 0, 0);
        else
          asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(side[g]) : "v"(a), "v"(b));
      }
      if (NOP) asm volatile("s_nop 15");
    }
    asm volatile("s_nop 15\n\ts_nop 15");  // MFMA results -> VALU reads below (inline asm hides the hazard)
    const f32x4 res = (ROT == 1 && (LINKS & 1)) ? alt : ((ROT == 1) ? acc : acc);
    const f32x4 fin = (ROT == 1) ? ((LINKS & 1) ? alt : acc) : res;
    const float want = 32.0f * LINKS;
    for (int j = 0; j < 4; ++j) wrong += (fin[j] != want);
    for (int g = 0; g < GAP; ++g)
      for (int j = 0; j < 4; ++j) wrong += (side[g][j] != want);
  }
  if (wrong) atomicAdd(bad, wrong);
}

// PARTIAL overlap of vDst with SrcC (the allocation hipcc produces when it "rotates" a chain by two
// registers: v_mfma v[8:11], a, b, v[10:13] ; v_mfma v[10:13], a, b, v[8:11] ; ...), fixed registers.
template <int NOP>
__global__ void __launch_bounds__(512) k_partial(int reps, unsigned long long* bad) {
  h8 a, b;
  for (int i = 0; i < 8; ++i) {
    a[i] = (_Float16)1.0f;
    b[i] = (_Float16)1.0f;
  }
  unsigned long long wrong = 0;
  for (int r = 0; r < reps; ++r) {
    for (int s = 0; s < ((threadIdx.x >> 6) + r) % 5; ++s) asm volatile("s_nop 7");
    float o0, o1, o2, o3;
    asm volatile(
        "v_mov_b32 v10, 0\n v_mov_b32 v11, 0\n v_mov_b32 v12, 0\n v_mov_b32 v13, 0\n s_nop 7\n"
        ".rept 32\n"
        "v_mfma_f32_16x16x32_f16 v[8:11], %4, %5, v[10:13]\n"
        ".if %6\n s_nop 15\n .endif\n"
        "v_mfma_f32_16x16x32_f16 v[10:13], %4, %5, v[8:11]\n"
        ".if %6\n s_nop 15\n .endif\n"
        ".endr\n"
        "s_nop 15\n s_nop 15\n"
        "v_mov_b32 %0, v10\n v_mov_b32 %1, v11\n v_mov_b32 %2, v12\n v_mov_b32 %3, v13\n"
        : "=v"(o0), "=v"(o1), "=v"(o2), "=v"(o3)
        : "v"(a), "v"(b), "n"(NOP)
        : "v8", "v9", "v10", "v11", "v12", "v13");
    const float want = 32.0f * 64;
    wrong += (o0 != want) + (o1 != want) + (o2 != want) + (o3 != want);
  }
  if (wrong) atomicAdd(bad, wrong);
}

template <int NOP>
static void run_partial(const char* name) {
  unsigned long long* bad;
  (void)hipMalloc(&bad, 8);
  (void)hipMemset(bad, 0, 8);
  const int reps = 400;
  hipLaunchKernelGGL((k_partial<NOP>), dim3(1024), dim3(512), 0, 0, reps, bad);
  (void)hipDeviceSynchronize();
  unsigned long long h = 0;
  (void)hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost);
  printf("%-58s wrong accumulator entries %12llu of %.3g\n", name, h, 1024.0 * 512 * reps * 4);
  (void)hipFree(bad);
}

template <int GAP, int ROT, int NOP>
static void run(const char* name) {
  unsigned long long* bad;
  (void)hipMalloc(&bad, 8);
  (void)hipMemset(bad, 0, 8);
  const int reps = 400;
  hipLaunchKernelGGL((k_chain<GAP, ROT, NOP>), dim3(1024), dim3(512), 0, 0, reps, bad);
  (void)hipDeviceSynchronize();
  unsigned long long h = 0;
  (void)hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost);
  const double total = 1024.0 * 512 * reps * 4 * (1 + GAP);
  printf("%-58s wrong accumulator entries %12llu of %.3g\n", name, h, total);
  (void)hipFree(bad);
}

int main() {
  run<0, 0, 0>("in place  (vDst == SrcC), back to back");
  run<1, 0, 0>("in place, 1 independent MFMA between links");
  run<2, 0, 0>("in place, 2 independent MFMAs between links");
  run<3, 0, 0>("in place, 3 independent MFMAs between links");
  run_partial<0>("PARTIAL overlap v[8:11] <- v[10:13] <- v[8:11], back to back");
  run_partial<1>("PARTIAL overlap, s_nop 15 between links");
  run<0, 2, 0>("intrinsic (compiler allocation), back to back");
  run<1, 2, 0>("intrinsic, 1 independent MFMA between links");
  run<2, 2, 0>("intrinsic, 2 independent MFMAs between links");
  run<3, 2, 0>("intrinsic, 3 independent MFMAs between links");
  return 0;
}
