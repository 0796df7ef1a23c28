// tools/stream_bench.hip - what the memory system delivers for the edge kernel's ACCESS PATTERN with
// no arithmetic at all: persistent 256-workgroup walk over tile-blocked edge latents (8 KiB per
// 16-edge tile read, 8 KiB written), optional psr gathers (2 x 512 B per edge from a 64 MB table),
// optional second read of the tile.  Variants: waves per SIMD, tiles in flight per wave, in place vs
// out of place, XCD-contiguous vs interleaved walk.  Sets the ceiling k_edge16v can be priced against.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/stream_bench.hip -o tools/bin/stream_bench
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

// MODE bit 0: gathers, bit 1: reload, bit 2: out-of-place store, bit 3: no store, bit 4: no e load,
// bit 5: interleaved walk (tile = wave_global + it * total_waves) instead of XCD-contiguous,
// bit 6: nontemporal stores, bit 7: nontemporal e loads, bit 8: gather only the sender half
template <int WPS, int TPW, int MODE>
__global__ void __launch_bounds__(WPS * 256, WPS) k_stream(const f32x4* __restrict__ src, f32x4* __restrict__ dst,
                                                           const f32x4* __restrict__ psr, const int* __restrict__ snd,
                                                           const int* __restrict__ rcv, int ntiles, int rev) {
  constexpr int WAVES = WPS * 4;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, g = lane >> 4;
  int t, stride, t_hi;
  if (MODE & 32) {
    t = blockIdx.x * WAVES + wave;
    stride = gridDim.x * WAVES;
    t_hi = ntiles;
  } else {
    const int xcd = blockIdx.x & 7, slot = (blockIdx.x >> 3) * WAVES + wave;
    stride = (gridDim.x >> 3) * WAVES;
    const int t_lo = (int)(((int64_t)ntiles * xcd) >> 3);
    t_hi = (int)(((int64_t)ntiles * (xcd + 1)) >> 3);
    t = t_lo + slot;
  }
  for (; t < t_hi; t += stride * TPW) {
    f32x4 v[TPW][8], p[TPW][8];
#pragma unroll
    for (int u = 0; u < TPW; ++u) {
      const int tf = min(t + u * stride, t_hi - 1);
      const int tt = rev ? ntiles - 1 - tf : tf;
      const f32x4* er = src + (int64_t)tt * 512 + lane;
#pragma unroll
      for (int mb = 0; mb < 8; ++mb)
        v[u][mb] = (MODE & 16) ? f32x4{1.f, (float)tt, 2.f, (float)mb}
                               : ((MODE & 128) ? __builtin_nontemporal_load(&er[64 * mb]) : er[64 * mb]);
      if (MODE & 1) {
        const int s = snd[tt * 16 + n], r = rcv[tt * 16 + n];
        const f32x4* ps = psr + (int64_t)s * 64 + g;
        const f32x4* pr = psr + (int64_t)r * 64 + 32 + g;
#pragma unroll
        for (int mb = 0; mb < 8; ++mb) p[u][mb] = (MODE & 256) ? ps[4 * mb] : ps[4 * mb] + pr[4 * mb];
      }
    }
#pragma unroll
    for (int u = 0; u < TPW; ++u) {
      const int tf = t + u * stride;
      if (tf >= t_hi) break;
      const int tt = rev ? ntiles - 1 - tf : tf;
      if (MODE & 1) {
#pragma unroll
        for (int mb = 0; mb < 8; ++mb) v[u][mb] = v[u][mb] + p[u][mb];
      }
      if (MODE & 2) {
        const f32x4* er = src + (int64_t)tt * 512 + lane;
#pragma unroll
        for (int mb = 0; mb < 8; ++mb) v[u][mb] = v[u][mb] * 0.5f + er[64 * mb];
      }
      f32x4* ew = ((MODE & 4) ? dst : const_cast<f32x4*>(src)) + (int64_t)tt * 512 + lane;
      if (!(MODE & 8)) {
#pragma unroll
        for (int mb = 0; mb < 8; ++mb) {
          if (MODE & 64)
            __builtin_nontemporal_store(v[u][mb] * 1.0001f, &ew[64 * mb]);
          else
            ew[64 * mb] = v[u][mb] * 1.0001f;
        }
      } else {
        float acc = 0;
#pragma unroll
        for (int mb = 0; mb < 8; ++mb) acc += v[u][mb][0] + v[u][mb][3];
        if (acc == 123.456f) ew[0] = v[u][0];
      }
    }
  }
}

template <typename F>
static float time_it(F launch, int iters) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  launch();
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0, 0);
  for (int i = 0; i < iters; ++i) launch();
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  return 1e3f * ms / iters;
}

int main(int argc, char** argv) {
  const int64_t E = argc > 1 ? atoll(argv[1]) : 1097008, N = argc > 2 ? atoll(argv[2]) : 64000;
  const int ntiles = (int)(E / 16);
  f32x4 *src, *dst, *psr;
  int *snd, *rcv;
  (void)hipMalloc(&src, E * 512);
  (void)hipMalloc(&dst, E * 512);
  (void)hipMalloc(&psr, N * 1024);
  (void)hipMalloc(&snd, E * 4);
  (void)hipMalloc(&rcv, E * 4);
  (void)hipMemset(src, 0, E * 512);
  (void)hipMemset(dst, 0, E * 512);
  (void)hipMemset(psr, 0, N * 1024);
  {
    std::vector<int> s(E), r(E);
    for (int64_t k = 0; k < E; ++k) {
      r[k] = (int)(k * N / E);
      s[k] = (int)((r[k] + (rand() % 400) - 200 + N) % N);
    }
    (void)hipMemcpy(snd, s.data(), E * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(rcv, r.data(), E * 4, hipMemcpyHostToDevice);
  }
  const double rw = (double)E * 1024;
#define RUN(name, W, T, M, bytes) RUNR(name, W, T, M, bytes, 0)
#define RUNR(name, W, T, M, bytes, ALT)                                                                        \
  do {                                                                                                     \
    int flip = 0;                                                                                          \
    auto l = [&] {                                                                                         \
      flip ^= (ALT);                                                                                       \
      hipLaunchKernelGGL((k_stream<W, T, M>), dim3(256), dim3(W * 256), 0, 0, src, dst, psr, snd, rcv, ntiles, flip); \
    };                                                                                                     \
    const float us = time_it(l, 100);                                                                      \
    printf("%-58s %7.1f us  %5.2f TB/s HBM\n", name, us, (bytes) / us * 1e-6);                              \
    fflush(stdout);                                                                                        \
  } while (0)
  for (int i = 0; i < 3; ++i) {
    auto l = [&] { hipLaunchKernelGGL((k_stream<4, 1, 0>), dim3(256), dim3(1024), 0, 0, src, dst, psr, snd, rcv, ntiles, 0); };
    (void)time_it(l, 500);
  }
  printf("E=%lld: read+write %.3f GB; copy ceiling 6.29 TB/s = %.0f us\n", (long long)E, rw * 1e-9, rw / 6.29e6);
  RUN("in place r+w                  3 waves/SIMD", 3, 1, 0, rw);
  RUN("in place r+w                  4 waves/SIMD", 4, 1, 0, rw);
  RUN("in place r+w                  2 waves/SIMD 2 tiles/wave", 2, 2, 0, rw);
  // boustrophedon: every other launch walks the tiles in reverse, so the tail the previous launch wrote
  // (still in the 256 MiB Infinity Cache) is what the next launch reads first
  RUNR("in place r+w, alternate direction    4 waves/SIMD", 4, 1, 0, rw, 1);
  RUNR("in place r+w nt both, alternate dir  4 waves/SIMD", 4, 1, 192, rw, 1);
  RUNR("in place r+w nt loads, alternate dir 4 waves/SIMD", 4, 1, 128, rw, 1);
  RUNR("in place r+w nt stores, alternate    4 waves/SIMD", 4, 1, 64, rw, 1);
  RUNR("r+w + gathers, alternate direction   3 waves/SIMD", 3, 1, 1, rw, 1);
  RUNR("r+w + gathers nt both, alternate dir 3 waves/SIMD", 3, 1, 193, rw, 1);
  RUNR("r+w + gathers 2 tiles/wave, alt dir  2 waves/SIMD", 2, 2, 1, rw, 1);
  RUNR("r+w + gathers nt loads, alt dir      3 waves/SIMD", 3, 1, 129, rw, 1);
  RUNR("r+w + gathers nt stores, alt dir     3 waves/SIMD", 3, 1, 65, rw, 1);
  RUNR("r+w + gathers nt loads, alt dir      2 waves/SIMD", 2, 1, 129, rw, 1);
  RUNR("r+w + gathers nt both                2 waves/SIMD", 2, 1, 193, rw, 0);
  RUNR("r+w + gathers 2 tiles/wave nt loads, alt dir 2 w/SIMD", 2, 2, 129, rw, 1);
  RUN("in place r+w, nt stores       4 waves/SIMD", 4, 1, 64, rw);
  RUN("in place r+w, nt loads        4 waves/SIMD", 4, 1, 128, rw);
  RUN("in place r+w, nt both         4 waves/SIMD", 4, 1, 192, rw);
  RUN("in place r+w, nt both         3 waves/SIMD", 3, 1, 192, rw);
  RUN("out of place r+w              4 waves/SIMD", 4, 1, 4, rw);
  RUN("out of place r+w, nt both     4 waves/SIMD", 4, 1, 196, rw);
  RUN("interleaved walk r+w          4 waves/SIMD", 4, 1, 32, rw);
  RUN("interleaved walk r+w nt both  4 waves/SIMD", 4, 1, 224, rw);
  RUN("read only                     4 waves/SIMD", 4, 1, 8, rw / 2);
  RUN("read only nt                  4 waves/SIMD", 4, 1, 136, rw / 2);
  RUN("write only                    4 waves/SIMD", 4, 1, 16, rw / 2);
  RUN("write only nt                 4 waves/SIMD", 4, 1, 80, rw / 2);
  RUN("r+w + psr gathers             3 waves/SIMD", 3, 1, 1, rw);
  RUN("r+w + psr gathers             4 waves/SIMD", 4, 1, 1, rw);
  RUN("r+w + psr gathers, nt both    4 waves/SIMD", 4, 1, 193, rw);
  RUN("r+w + psr gathers, nt both    3 waves/SIMD", 3, 1, 193, rw);
  RUN("r+w + psr gathers, nt stores  4 waves/SIMD", 4, 1, 65, rw);
  RUN("r+w + sender gather only      4 waves/SIMD", 4, 1, 257, rw);
  RUN("r+w + sender gather only, nt  4 waves/SIMD", 4, 1, 449, rw);
  RUN("r+w + gathers + reload        4 waves/SIMD", 4, 1, 3, rw);
  RUN("r+w + gathers + reload, nt st 4 waves/SIMD", 4, 1, 67, rw);
  RUN("r+w + gathers 2 tiles/wave    2 waves/SIMD", 2, 2, 1, rw);
  RUN("r+w + gathers 2 tiles/wave nt 2 waves/SIMD", 2, 2, 193, rw);
  RUN("gathers only                  4 waves/SIMD", 4, 1, 25, rw);
  RUN("sender gather only            4 waves/SIMD", 4, 1, 281, rw);
  return 0;
}
