// tools/wstream_bench.hip - the node kernel's WEIGHT STREAM alone (k_node16s: 320 KiB of fp16 hi | lo fragments per workgroup in
// ten 32 KiB chunks through a two-slot LDS ring, one barrier per chunk, 500 workgroups of 8 waves = two per CU, every workgroup
// reads the SAME 320 KiB from L2), by two routes:
//   mode 0: global_load_lds_dwordx4 (direct to LDS, what the kernel ships)
//   mode 1: global_load_dwordx4 into registers (four 16-byte loads per lane and chunk), ds_write_b128 one step later
//   mode 2: mode 1 with the loads of chunk c + 2 in flight (two register sets)
// Each step every wave reads `reads` fragments of the current slot from LDS (the GEMM's operand traffic: 32 per chunk in the
// kernel) and folds them into a checksum, so that the slot is consumed.
// profiles/r02_node16s_ablation.txt measured "the weight stream alone" at 18 us of the node launch's 48 and found that nothing
// overlaps it; this tool asks whether the direct-to-LDS route is what makes it that long.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/wstream_bench.hip -o tools/bin/wstream_bench
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) f32x4* lds_ptr;
#define CHUNK 2048   // f32x4 per chunk (32 KiB)
#define NCH 10

#define CK(x)                                                                  \
  do {                                                                         \
    hipError_t e_ = (x);                                                       \
    if (e_ != hipSuccess) {                                                    \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                  \
      exit(1);                                                                 \
    }                                                                          \
  } while (0)

// PF: every workgroup touches one dword of every 128-byte line of the 320 KiB first (2560 lines: 5 or 2.5 loads per thread) -
// with cold weights (the rollout: 1 GB of edge traffic between two uses of a layer's matrices) the ten chunks are then ten
// L2 hits instead of ten dependent misses, each only one barrier step ahead of its use
template <int MODE, int NW, bool PF>
__global__ void __launch_bounds__(NW * 64, NW == 8 ? 4 : 2) k_wstream(const f32x4* __restrict__ w, float* __restrict__ out, int reads) {
  __shared__ f32x4 sB[2][CHUNK];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int PPW = 32 / NW;   // 1 KiB pieces per wave and chunk
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if constexpr (PF) {
    const float* wf = reinterpret_cast<const float*>(w);
    float t = 0.f;
    for (int l = tid; l < NCH * CHUNK / 8; l += NW * 64) t += wf[l * 32];
    acc[0] = t;
  }
  if constexpr (MODE == 0) {
    auto issue = [&](int c) {
      const f32x4* src = w + (size_t)c * CHUNK;
      f32x4* slot = sB[c & 1];
#pragma unroll
      for (int i = 0; i < PPW; ++i) {
        const int piece = wave + NW * i;
        const uint32_t voff = (uint32_t)(piece * 64 + lane) * 16u;
        const uint32_t lo = (uint32_t)(uintptr_t)(lds_ptr)(slot + piece * 64);
        asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(src), "s"(lo) : "memory");
      }
    };
    issue(0);
    issue(1);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      if (c == 0) {   // chunks 0 and 1 are out; later steps have only their own refill outstanding (the kernel's NS_STEP)
        if constexpr (PPW == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        if constexpr (PPW == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      if (c >= 1 && c + 1 < NCH) issue(c + 1);   // the slot chunk c - 1 released
      const f32x4* s = sB[c & 1] + lane;
      for (int r = 0; r < reads; ++r) acc = acc + s[(r * 64) & (CHUNK - 1)];
    }
  } else {
    constexpr int SETS = MODE == 2 ? 2 : 1;
    f32x4 st[SETS][PPW];
    auto fetch = [&](int c, f32x4 (&v)[PPW]) {
      const f32x4* src = w + (size_t)c * CHUNK;
#pragma unroll
      for (int i = 0; i < PPW; ++i) v[i] = src[(wave + NW * i) * 64 + lane];
    };
    auto put = [&](int c, const f32x4 (&v)[PPW]) {
      f32x4* slot = sB[c & 1];
#pragma unroll
      for (int i = 0; i < PPW; ++i) slot[(wave + NW * i) * 64 + lane] = v[i];
    };
    // registers hold chunks c + 1 .. c + SETS during step c: chunk k lives in set k % SETS
    fetch(0, st[0]);
    put(0, st[0]);
#pragma unroll
    for (int k = 1; k <= SETS; ++k) fetch(k, st[k % SETS]);
    __syncthreads();
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      // chunk c is in its slot; the slot of chunk c + 1 was released by the barrier behind step c - 1
      const f32x4* s = sB[c & 1] + lane;
      for (int r = 0; r < reads; ++r) acc = acc + s[(r * 64) & (CHUNK - 1)];
      if (c + 1 < NCH) {
        put(c + 1, st[(c + 1) % SETS]);
        if (c + 1 + SETS < NCH) fetch(c + 1 + SETS, st[(c + 1) % SETS]);
      }
      __syncthreads();
    }
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) out[blockIdx.x] = acc[0];
}

__global__ void k_thrash(const f32x4* __restrict__ src, f32x4* __restrict__ dst, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i] + 1.f;
}
// cold: a 512 MB copy between two launches (L2 and most of the Infinity Cache turn over), events around the stream launch only
template <int MODE, int NW, bool PF>
static float run_cold(const f32x4* w, float* out, int nblk, int reads, int reps, const f32x4* big, f32x4* big2, size_t nbig) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  float tot = 0.f;
  for (int i = 0; i < reps + 2; ++i) {
    hipLaunchKernelGGL(k_thrash, dim3(2048), dim3(256), 0, 0, big, big2, nbig);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_wstream<MODE, NW, PF>), dim3(nblk), dim3(NW * 64), 0, 0, w, out, reads);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (i >= 2) tot += ms;
  }
  return 1e3f * tot / reps;
}
template <int MODE, int NW>
static float run(const f32x4* w, float* out, int nblk, int reads, int reps) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k_wstream<MODE, NW, false>), dim3(nblk), dim3(NW * 64), 0, 0, w, out, reads);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k_wstream<MODE, NW, false>), dim3(nblk), dim3(NW * 64), 0, 0, w, out, reads);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return 1e3f * ms / reps;
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 200;
  f32x4* w;
  float* out;
  CK(hipMalloc(&w, (size_t)NCH * CHUNK * sizeof(f32x4)));
  CK(hipMalloc(&out, 4096 * sizeof(float)));
  std::vector<float> h((size_t)NCH * CHUNK * 4, 1.0f);
  CK(hipMemcpy(w, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
  for (int reads : {0, 32}) {
    printf("reads per chunk and lane: %d   (us per launch)\n", reads);
    printf("  8 waves x 500 workgroups (two per CU):  direct-to-LDS %.2f   registers %.2f   registers, 2 ahead %.2f\n",
           run<0, 8>(w, out, 500, reads, reps), run<1, 8>(w, out, 500, reads, reps), run<2, 8>(w, out, 500, reads, reps));
    printf("  16 waves x 250 workgroups (one per CU): direct-to-LDS %.2f   registers %.2f   registers, 2 ahead %.2f\n",
           run<0, 16>(w, out, 250, reads, reps), run<1, 16>(w, out, 250, reads, reps), run<2, 16>(w, out, 250, reads, reps));
  }
  const size_t nbig = (size_t)32 << 20;   // f32x4: 512 MB
  f32x4 *big, *big2;
  CK(hipMalloc(&big, nbig * sizeof(f32x4)));
  CK(hipMalloc(&big2, nbig * sizeof(f32x4)));
  CK(hipMemset(big, 0, nbig * sizeof(f32x4)));
  const int rc = reps < 40 ? reps : 40;
  printf("COLD weights (512 MB copied between launches; events around the stream launch; 32 reads):\n");
  printf("  8 waves x 500:  direct-to-LDS %.2f   + touch-all-lines first %.2f   registers %.2f   + touch first %.2f\n",
         run_cold<0, 8, false>(w, out, 500, 32, rc, big, big2, nbig), run_cold<0, 8, true>(w, out, 500, 32, rc, big, big2, nbig),
         run_cold<1, 8, false>(w, out, 500, 32, rc, big, big2, nbig), run_cold<1, 8, true>(w, out, 500, 32, rc, big, big2, nbig));
  printf("  (empty launch between the same events: %.2f)\n", run_cold<0, 8, false>(w, out, 0 + 1, 0, rc, big, big2, nbig));
  return 0;
}
