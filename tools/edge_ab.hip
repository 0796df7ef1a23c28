// tools/edge_ab.hip - A/B micro-benchmark of the product edge kernel k_edge16v (2 waves per SIMD, resident latents,
// GEMM-phase priority, nontemporal streams) on a synthetic receiver-sorted edge list of the TGV3D-8k x 8 size
// (E = 1.097 M, N = 64 k); built twice, with -DLB_GEMM_INTERLEAVE=0 (round 3's block loop: the operand split as one
// 12-instruction block between MFMA phases) and =1 (round 4: one split instruction per MFMA slot):
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -Ilagrangebench_amd/csrc [-DLB_GEMM_INTERLEAVE=0] tools/edge_ab.hip -o tools/bin/edge_ab_i{0,1}
// Timing: warm clocks (as many untimed launches as timed ones first - after an idle gap the first ~50 ms run 15-20 % slow).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../lagrangebench_amd/csrc/lb_edge16.hip"
#include "../lagrangebench_amd/csrc/lb_edge16v.hip"

thread_local std::string g_lb_err;
int lb_fail(int code, const char*, ...) { return code; }

template <typename F>
static float time_it(F launch, int iters) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  for (int i = 0; i < 20; ++i) launch();  // (the clocks are warm: main() runs 6 s of launches first)
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
  const int64_t E = argc > 1 ? atoll(argv[1]) : 1097000, N = argc > 2 ? atoll(argv[2]) : 64000;
  const int iters = argc > 3 ? atoi(argv[3]) : 200;
  std::vector<int> s(E), r(E), rp(N + 1);
  srand(1);
  for (int64_t k = 0; k < E; ++k) {
    r[k] = (int)(k * N / E);
    s[k] = (int)((r[k] + (rand() % 400) - 200 + N) % N);
  }
  {
    int64_t k = 0;
    rp[0] = 0;
    for (int64_t g = 0; g < N; ++g) {
      while (k < E && r[k] == g) ++k;
      rp[g + 1] = (int)k;
    }
  }
  lb_ctrl c{};
  c.overflow_step = -1;
  c.n_edges_total = (int)E;
  lb_ctrl* dc;
  int *ds, *dr, *drp;
  float *psr, *w, *vec;
  struct { int64_t E, N; float *elat, *elat0, *agg, *part; } b{};
  b.E = E;
  b.N = N;
  (void)hipMalloc(&dc, sizeof(c));
  (void)hipMemcpy(dc, &c, sizeof(c), hipMemcpyHostToDevice);
  (void)hipMalloc(&ds, E * 4);
  (void)hipMalloc(&dr, E * 4);
  (void)hipMalloc(&drp, (N + 1) * 4);
  (void)hipMemcpy(ds, s.data(), E * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(dr, r.data(), E * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(drp, rp.data(), (N + 1) * 4, hipMemcpyHostToDevice);
  (void)hipMalloc(&b.elat, (E + 32) * 512);
  (void)hipMalloc(&b.elat0, (E + 32) * 512);
  (void)hipMalloc(&psr, N * 1024);
  (void)hipMalloc(&b.agg, N * 512);
  (void)hipMalloc(&b.part, (E / 16 + 2) * 1024);
  (void)hipMalloc(&w, 2 * 65536);
  (void)hipMalloc(&vec, 4 * 512);
  {
    std::vector<float> h((size_t)(E + 32) * 128);  // incl. the padding rows of the last (blocked) tile
    for (auto& x : h) x = (rand() % 2001 - 1000) * 1e-3f;
    (void)hipMemcpy(b.elat0, h.data(), (E + 32) * 512, hipMemcpyHostToDevice);
    (void)hipMemcpy(psr, h.data(), N * 1024 < E * 512 ? N * 1024 : E * 512, hipMemcpyHostToDevice);
    std::vector<float> hw(32768), packed(32768), hv(512);
    for (auto& x : hw) x = (rand() % 2001 - 1000) * 1e-4f;
    for (auto& x : hv) x = 0.5f + (rand() % 1001) * 1e-3f;
    // LayerNorm scale 1e-3, offset 0: the latents are updated IN PLACE, e += LayerNorm(..), and with an O(1) increment per
    // launch they leave the fp16 range within a measurement - every wave's range probe then raises the LARGE flag with a
    // same-address atomic and the launch reads 350 us instead of 250 (the flags are printed per line: they must read 0)
    for (int i = 128; i < 256; ++i) hv[i] *= 1e-3f;
    for (int i = 256; i < 384; ++i) hv[i] = 0.f;
    lb_pack_weight16h(hw.data(), 128, 128, 128, packed.data(), 128);
    lb_pack_weight16h(hw.data() + 16384, 128, 128, 128, packed.data() + 16384, 128);
    (void)hipMemcpy(w, packed.data(), 2 * 65536, hipMemcpyHostToDevice);
    (void)hipMemcpy(vec, hv.data(), 4 * 512, hipMemcpyHostToDevice);
  }
  lb_edge16_args a{};
  a.ctrl = dc;
  a.senders = ds;
  a.receivers = dr;
  a.elat = b.elat;
  a.psr = psr;
  a.w0p = w;
  a.w1p = w + 16384;
  a.b1 = vec;
  a.ln_s = vec + 128;
  a.ln_o = vec + 256;
  a.b0 = vec + 384;
  a.fused = 1;
  a.row_ptr = drp;
  a.agg = b.agg;
  a.part = b.part;
  const double bytes = (double)E * 1032 + (double)N * 1536;
  printf("E=%lld N=%lld algorithmic bytes %.3f GB (8 TB/s: %.0f us, 6.29 TB/s copy ceiling: %.0f us)\n", (long long)E,
         (long long)N, bytes * 1e-9, bytes / 8e6, bytes / 6.29e6);

  (void)hipMemcpy(b.elat, b.elat0, (E + 32) * 512, hipMemcpyDeviceToDevice);
  auto report = [&](const char* name, auto launch) {
    (void)hipMemcpy(b.elat, b.elat0, (E + 32) * 512, hipMemcpyDeviceToDevice);
    (void)hipMemcpy(dc, &c, sizeof(c), hipMemcpyHostToDevice);  // range-guard flags cleared
    const float us = time_it(launch, iters);
    lb_ctrl cc{};
    (void)hipMemcpy(&cc, dc, sizeof(cc), hipMemcpyDeviceToHost);
    printf("%-64s %8.1f us  %5.2f TB/s  frac(8TB/s) %.3f  guard flags %d\n", name, us, bytes / us * 1e-6, bytes / us * 1e-6 / 8.0,
           cc.math_flags);
    fflush(stdout);
  };
  printf("LB_GEMM_INTERLEAVE=%d\n", (int)LB_GEMM_INTERLEAVE);
  {  // the GPU takes SECONDS of sustained work after process start to reach its steady clock (the first measurements of a
     // process read 350 us for a kernel that runs 255 us five seconds later): 6 s of the product kernel first
    // (the SKIP variant: it does not store the latents.  The kernel updates them IN PLACE, e += LayerNorm(..): over
    // thousands of launches they outgrow the fp16 range and every wave's range probe then raises the guard flag with a
    // same-address atomic - 350 us per launch instead of 255.  Every measurement below restarts from the same latents.)
    auto k = [&] { hipLaunchKernelGGL((k_edge16v<2, false, true, 0, true, true, 1, false>), dim3(256), dim3(512), 0, 0, a); };
    for (int rep = 0; rep < 20; ++rep) {
      for (int i = 0; i < 1000; ++i) k();
      (void)hipDeviceSynchronize();
    }
  }
  for (int round = 0; round < 2; ++round) {
#define K(SKIP, ABL, G) [&] { hipLaunchKernelGGL((k_edge16v<2, false, SKIP, ABL, true, true, G, false>), dim3(256), dim3(512), 0, 0, a); }
  report("k_edge16v product (guard rows)", K(false, 0, 1));
  report("k_edge16v product (no guard)", K(false, 0, 0));
  report("k_edge16v last layer (no edge-latent store)", K(true, 0, 1));
  report("  compute only (no loads, no stores)", K(false, 7, 1));
  report("  no GEMMs (memory + VALU)", K(false, 8, 1));
  report("  no LayerNorm / scan", K(false, 16, 1));
  report("k_edge16v product (guard rows) AGAIN", K(false, 0, 1));
  {  // out of place: the latents do not evolve from launch to launch
    static float* other = nullptr;
    if (!other) (void)hipMalloc(&other, (E + 32) * 512);
    a.elat_out = other;
    report("k_edge16v product, out of place", K(false, 0, 1));
    a.elat_out = nullptr;
  }
#define K3(SKIP, ABL, G) [&] { hipLaunchKernelGGL((k_edge16v<3, false, SKIP, ABL, true, true, G, false>), dim3(256), dim3(768), 0, 0, a); }
  report("k_edge16v 3 waves per SIMD", K3(false, 0, 1));
  report("  3 waves, compute only", K3(false, 7, 1));
  }
  return 0;
}
